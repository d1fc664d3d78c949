"""GPU parity of ResUnet (archs/ResUnet.py:3-88) against the golden vectors captured from the
reference module and the torch-fp32 oracle: forward, loss, gradients (autograd path) and three
fused train steps."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _probe(a, idx):
    return np.asarray(a, np.float32).reshape(-1)[idx]


@pytest.mark.parametrize('res', [False, True])
def test_resunet_nf8_golden(golden_dir, res):
    from oracle import net_torch as O
    from pnnp_amd.archs import ResUnet
    from pnnp_amd.trainer import HipTrainStep
    g = np.load(os.path.join(golden_dir, f'resunet_nf8_res{int(res)}.npz'))
    sd = O.init_state(O.resunet_param_shapes(nf=8), seed=42)
    net = ResUnet(dict(nframes=1, res=res, nf=8, in_nc=4, out_nc=4))
    net.load_state_dict({k: v.clone() for k, v in sd.items()})
    net = net.cuda()
    x = torch.from_numpy(g['x']).cuda(); t = torch.from_numpy(g['t']).cuda()
    with torch.no_grad():
        y0 = net(x)
    np.testing.assert_allclose(y0.cpu().numpy(), g['y'], rtol=1e-4, atol=2e-6)
    y = net(x)
    loss = torch.nn.functional.l1_loss(y.clamp(0, 1), t)
    assert abs(loss.item() - float(g['loss'])) < 1e-6
    loss.backward()
    for k, p in net.named_parameters():
        got = _probe(p.grad.cpu().numpy(), g['g:' + k + ':idx'])
        ref = g['g:' + k + ':val']
        tol = 1e-3 * np.abs(ref).max() + 1e-9
        assert np.abs(got - ref).max() <= tol, (k, np.abs(got - ref).max(), tol)
        assert abs(float(p.grad.double().norm()) - g['g:' + k + ':sum'][1]) <= 1e-3 * g['g:' + k + ':sum'][1] + 1e-9, k
    # fused train step (flat buffers, fused Adam): 3 steps vs the reference's losses
    net2 = ResUnet(dict(nframes=1, res=res, nf=8, in_nc=4, out_nc=4))
    net2.load_state_dict({k: v.clone() for k, v in sd.items()})
    net2 = net2.cuda()
    ts = HipTrainStep(net2, lr=1e-4, clip=0)
    for it in range(3):
        lo = ts.step(t, noisy=x)
        assert abs(float(lo[0]) - g['train_losses'][it, 0]) < 5e-6, (it, float(lo[0]), g['train_losses'][it, 0])
    for k, p in net2.named_parameters():
        got = _probe(p.detach().cpu().numpy(), np.linspace(0, p.numel() - 1, min(32, p.numel())).astype(np.int64))
        np.testing.assert_allclose(got, g['w3:' + k + ':val'], rtol=2e-3, atol=5e-6, err_msg=k)


def test_resunet_nf32_full_crop_golden(golden_dir):
    from oracle import net_torch as O
    from pnnp_amd.archs import ResUnet
    g = np.load(os.path.join(golden_dir, 'resunet_nf32_512.npz'))
    sd = O.init_state(O.resunet_param_shapes(nf=32), seed=7)
    net = ResUnet(dict(nframes=1, res=False, nf=32, in_nc=4, out_nc=4))
    net.load_state_dict({k: v.clone() for k, v in sd.items()})
    net = net.cuda().eval()
    x = torch.rand(1, 4, 512, 512, generator=torch.Generator().manual_seed(0)).cuda()
    with torch.no_grad():
        y = net(x)
    np.testing.assert_allclose(_probe(y.cpu().numpy(), g['idx']), g['val'], rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(y.double().sum(dim=(0, 2, 3)).cpu().numpy(), g['chan_sum'], rtol=1e-4)


def test_resunet_nf32_grads_vs_oracle():
    from oracle import net_torch as O
    from pnnp_amd.archs import ResUnet, initialize_weights
    from pnnp_amd.trainer import HipTrainStep
    torch.manual_seed(5)
    net = ResUnet(dict(nframes=1, res=False, nf=32, in_nc=4, out_nc=4))
    initialize_weights(net)
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    net = net.cuda()
    x = torch.rand(1, 4, 96, 128); t = torch.rand(1, 4, 96, 128)
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    loss_ref = O.l1_clamp_loss(O.resunet_forward(leaves, x), t)
    loss_ref.backward()
    ts = HipTrainStep(net, lr=0.0, clip=0)
    lo = ts.step(t.cuda(), noisy=x.cuda())
    assert abs(float(lo[0]) - loss_ref.item()) < 2e-6
    for k, p in net.named_parameters():
        got = net.engine.params.grad_view(k, p.shape).cpu()
        ref = leaves[k].grad
        rel = float((got - ref).norm() / (ref.norm() + 1e-12))
        assert rel < 2e-3, (k, rel)
