"""Pin oracle/noiseflow_torch.py to the reference's NoiseFlow.sample (injected z): bit-exact on CPU."""
import os

import numpy as np
import torch

from oracle import noiseflow_torch as N


def test_sample_matches_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, 'noiseflow.npz'))
    sd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith('sd:')}
    assert len(sd) == 222
    for iso in (100, 1600, 3000, 6400):
        x = N.sample(sd, torch.from_numpy(g['clean']), torch.tensor(float(iso)), torch.from_numpy(g['z']))
        assert np.array_equal(x.numpy(), g[f'out_iso{iso}']), iso


def test_product_module_state_dict_contract(golden_dir):
    from pnnp_amd.archs import NoiseFlow
    g = np.load(os.path.join(golden_dir, 'noiseflow.npz'))
    net = NoiseFlow({'x_shape': (4, 32, 32), 'arch': 'sdn|unc|unc|unc|unc|giso|unc|unc|unc|unc'})
    assert list(net.state_dict().keys()) == list(g['keys'])
    for k, v in net.state_dict().items():
        assert tuple(v.shape) == g['sd:' + k].shape, k
