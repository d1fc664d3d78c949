"""Running a persistent convolution grid NEXT to another resident kernel (gradient all-reduce
overlapped with the backward pass, base_trainer.py:115-118).  The 1-GPU boxes cannot run RCCL with two ranks, so a kernel that
occupies k CUs stands in for the collective (tools/ubench/squat.hip)."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))

pytestmark = pytest.mark.gpu


def test_persistent_split_is_bit_identical_and_reports_the_cost_of_occupied_cus():
    """With one workgroup per CU and static shares, the workgroups whose CUs are occupied start when the others have FINISHED and the
    layer takes about twice as long.  With pnnp_set_persistent_split(4) the hardware dispatcher hands the quarter shares to whichever
    CU frees up.  The HARD check is the contract: the split changes which workgroup runs a tile, never a result bit.  The wall-clock
    ratios depend on the dispatcher, clocks, the power state and box noise, so they are printed (and kept per round in
    profiles/r*/squat_test.txt from `python tools/squat_test.py 32`: measured x 1.38-1.50 beside a kernel on 32 CUs with static shares,
    x 0.99-1.08 with quarter shares, 6-14 % for the split alone on the chip).  The SOFT check that the mechanism still helps:
    beside the squatter the quarter shares must be at least 5 % faster than the static ones (measured: 25-30 %; medians of 5 repetitions).
    The occupying kernel (tools/ubench/libsquat.so) is built by tools/build.py with the library -- never from this process, which has
    initialised the GPU by the time the suite gets here."""
    import squat_test
    try:
        r = squat_test.measure(32, may_build=False)
    except FileNotFoundError as e:
        pytest.skip(f'{e} not built (python tools/build.py)')
    print(r)
    assert r['same_result']
    # the soft check only where the hazard it is about was actually observed in this run (the occupying kernel resident before the layer was dispatched:
    # static shares beside it >= 1.2 x alone); a run in which the squatter came late measures nothing and must not fail the suite
    if r['beside_1'] >= 1.2 * r['alone_1']:
        assert r['beside_4'] <= 0.95 * r['beside_1'], r
    else:
        print('squat_test: the occupying kernel was not resident in time (static shares beside it %.3f ms vs alone %.3f ms): ratio not checked' % (r['beside_1'], r['alone_1']))
    assert r['beside_4'] < 1.5 * r['beside_1'], r


def test_persistent_split_covers_the_weight_gradient_too():
    """Round 5: with pnnp_set_persistent_split(4) the 3x3 backward-weight kernels launch four quarter-share workgroups
    per CU as well (four times the slabs, reduced by slab INDEX in a fixed order), so an all-reduce overlapping any part of the backward pass finds
    dynamic shares.  Contract: deterministic for a given share (two runs bit-identical), and the same sums as with static shares up to the rounding
    of another pixel partition; both kernel families (fp16x2, bf16x3)."""
    from pnnp_amd import ops
    B, S, Ci, Co = 4, 128, 64, 64
    g = torch.Generator(device='cuda').manual_seed(3)
    x = torch.randn(B, S, S, Ci, device='cuda', generator=g); gy = torch.randn(B, S, S, Co, device='cuda', generator=g)
    ws = torch.empty(ops.x3_wgrad_workspace_floats(B, S, S, Co, Ci), device='cuda')
    slot = lambda t: ops.amax(t, torch.zeros(1, dtype=torch.int32, device='cuda'))
    sg, sx = slot(gy), slot(x)
    def run(split, h2):
        ops.set_persistent_split(split)
        try:
            dW = torch.empty(Co, Ci, 3, 3, device='cuda'); db = torch.empty(Co, device='cuda')
            if h2:
                ops.conv_h2_bwd_weight(gy, sg, Co, x, sx, Ci, None, None, dW, db, ws)
            else:
                ops.conv_x3_bwd_weight(gy, Co, x, Ci, None, dW, db, ws)
            return dW, db
        finally:
            ops.set_persistent_split(1)
    for h2 in (True, False):
        w1, b1 = run(1, h2); w4, b4 = run(4, h2); w4b, b4b = run(4, h2)
        assert torch.equal(w4, w4b) and torch.equal(b4, b4b)                                   # deterministic
        assert float((w4 - w1).abs().max()) <= 2e-5 * float(w1.abs().max()) and not torch.equal(w4, w1)      # another partition: rounding only
        assert float((b4 - b1).abs().max()) <= 2e-5 * float(b1.abs().max())
