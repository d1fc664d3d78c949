"""Host logic of the kernel-family policy (no GPU): ADVICE round 5 -- explicit h2_wgrad / h2_pointwise overrides survive later set_policy calls."""
from pnnp_amd.archs.unet import _EngineBase


def test_set_policy_keeps_explicit_h2_sub_switches():
    e = _EngineBase()
    e._init_base()
    assert e.policy.h2 and e.policy.h2_wgrad and e.policy.h2_pointwise
    e.set_policy(h2_wgrad=False)
    assert not e.policy.h2_wgrad and e.policy.h2_pointwise
    e.set_policy(x3=True)                       # a later call that does not name it: the override stays
    assert not e.policy.h2_wgrad and e.policy.x3
    e.set_policy(h2_pointwise=False)            # accepted as a keyword (it used to raise TypeError)
    assert not e.policy.h2_pointwise and not e.policy.h2_wgrad
    e.set_policy(h2=False)                      # sub-switches follow the family switch
    assert not e.policy.h2_wgrad and not e.policy.h2_pointwise
    e.set_policy(h2=True, h2_wgrad=True)
    assert e.policy.h2_wgrad and not e.policy.h2_pointwise
    assert len(e.policy.key()) == 12
