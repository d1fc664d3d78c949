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


def test_forward_and_loss_match_reference(golden_dir):
    """Density direction (row f4): z, the log-det objective and the NLL of the reference's forward()/loss()."""
    import os
    import numpy as np
    import torch
    from oracle import noiseflow_torch as O
    g = np.load(os.path.join(golden_dir, 'noiseflow.npz'), allow_pickle=False)
    sd = {k: torch.from_numpy(g['sd:' + k]) for k in [str(x) for x in g['keys']]}
    noise, clean = torch.from_numpy(g['fw_noise']), torch.from_numpy(g['clean'])
    for iso in (100, 3000):
        z, obj = O.forward(sd, noise, clean, torch.tensor(float(iso)))
        assert torch.allclose(z, torch.from_numpy(g[f'fw_z_iso{iso}']), rtol=1e-5, atol=1e-6)
        assert torch.allclose(obj, torch.from_numpy(g[f'fw_obj_iso{iso}']), rtol=1e-6)
        nll, sdz = O.loss(sd, noise, clean, torch.tensor(float(iso)))
        assert abs(float(nll) - g[f'fw_nll_iso{iso}'][0]) < 1e-5 * abs(g[f'fw_nll_iso{iso}'][0])
        assert abs(float(sdz) - g[f'fw_nll_iso{iso}'][1]) < 1e-7


def test_train_step_grads_match_reference(golden_dir):
    """Fitting (row f4): one train-mode loss().backward() of the reference -> NLL, every trainable parameter's gradient,
    the BatchNorm running statistics after the step.  Same torch ops on the same CPU: agreement to float rounding."""
    from oracle import noiseflow_torch as O
    g = np.load(os.path.join(golden_dir, 'noiseflow.npz'), allow_pickle=False)
    sd = {k: torch.from_numpy(g['sd:' + k]) for k in [str(x) for x in g['keys']]}
    noise, clean = torch.from_numpy(g['tr_noise']), torch.from_numpy(g['tr_clean'])
    for iso in (1600, 3000):
        nll, sdz, grads, bufs = O.loss_and_grads(sd, noise, clean, torch.tensor(float(iso)))
        assert abs(float(nll) - g[f'tr_nll_iso{iso}'][0]) < 1e-6 * abs(g[f'tr_nll_iso{iso}'][0])
        names = [k.split(':', 1)[1] for k in g.files if k.startswith(f'tr_grad_iso{iso}:')]
        assert sorted(names) == sorted(grads.keys()) and len(names) == 8 * 3 + 8 * 12 + 3 + 2
        for k in names:
            ref = g[f'tr_grad_iso{iso}:' + k]
            tol = 1e-5 * max(1e-12, np.abs(ref).max()) + 1e-9
            assert np.abs(grads[k].numpy() - ref).max() <= tol, (k, np.abs(grads[k].numpy() - ref).max(), np.abs(ref).max())
        for k, v in bufs.items():
            np.testing.assert_allclose(v.numpy(), g[f'tr_buf_iso{iso}:' + k], rtol=1e-6, atol=1e-8, err_msg=k)
