"""GPU parity of the dataset-side crop/augment kernels (rows f2, f3) against goldens produced by the
reference's dataset classes (tests/golden/make_golden.py augment).  Bit-exact."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def G(golden_dir):
    return np.load(os.path.join(golden_dir, 'augment.npz')), json.load(open(os.path.join(golden_dir, 'augment.json')))


def _crop(meta, tag, ways, H=160, W=224):
    from pnnp_amd.augment import CropAugment
    m = meta[tag]
    ca = CropAugment({'H': H, 'W': W, 'patch_size': m['ps'], 'crop_per_image': m['crop_per_image']}, ways=ways)
    np.random.seed(m['seed'])
    ca.init_random_crop_point(mode=m['mode'])
    assert [int(v) for v in ca.h_start] == m['h_start'] and [int(v) for v in ca.w_start] == m['w_start']
    assert [int(v) for v in ca.aug] == m['aug']
    return ca


@pytest.mark.parametrize('tag,ways', [('syn_random', 8), ('syn_grid', 8), ('real_random', 4)])
def test_fused_crop_pack_matches_reference(G, tag, ways):
    g, meta = G
    ca = _crop(meta, tag, ways)
    frame = torch.from_numpy(g['frame']).cuda()
    out = ca.crop_pack(frame, wp=16383, bl=512, norm=True, clip=True)
    assert np.array_equal(out.cpu().numpy(), g[tag + '_crops'])
    # and the two-step form: pack the frame, then random_crop the packed image
    from pnnp_amd.isp_ops import raw2bayer
    hr = raw2bayer(frame, wp=16383, bl=512, norm=True, clip=True)
    assert np.array_equal(ca.random_crop(hr).cpu().numpy(), g[tag + '_crops'])
    # numpy in -> numpy out, like a DataLoader worker would call it
    assert np.array_equal(ca.random_crop(hr.cpu().numpy()), g[tag + '_crops'])


def test_every_aug_mode(G):
    from pnnp_amd.augment import CropAugment
    from pnnp_amd.isp_ops import raw2bayer
    g, _ = G
    hr = raw2bayer(torch.from_numpy(g['frame']).cuda(), wp=16383, bl=512, norm=True, clip=False)
    crop = hr[:, 3:39, 7:43].contiguous()
    c8 = CropAugment({'H': 160, 'W': 224, 'patch_size': 36, 'crop_per_image': 8}, ways=8)
    for m in range(8):
        assert np.array_equal(c8.data_aug(crop, mode=m).cpu().numpy(), g['aug8'][m]), m
    c4 = CropAugment({'H': 160, 'W': 224, 'patch_size': 36, 'crop_per_image': 4}, ways=4)
    for m in range(4):
        assert np.array_equal(c4.data_aug(crop, mode=m).cpu().numpy(), g['aug4'][m]), m


def test_wb_gains(G):
    from pnnp_amd.augment import random_gains
    g, meta = G
    mg = meta['gains']
    torch.manual_seed(mg['torch_seed']); np.random.seed(mg['np_seed'])
    rgb, red, blue = random_gains()
    assert float(rgb[0]) == mg['rgb'] and float(red[0]) == mg['red_raw'] and float(blue[0]) == mg['blue_raw']
    wb = np.array(mg['wb'], np.float32)
    red_g, blue_g = wb[0] / red.numpy(), wb[2] / blue.numpy()
    assert float(red_g[0]) == mg['red'] and float(blue_g[0]) == mg['blue']
    ca = _crop(meta, 'syn_random', 8)
    frame = torch.from_numpy(g['frame']).cuda()
    out = ca.crop_pack(frame, wp=16383, bl=512, norm=True, clip=True, gains=(rgb.numpy(), red_g, blue_g))
    assert np.array_equal(out.cpu().numpy(), g['gain_crops'])
    out = ca.crop_pack(frame, wp=16383, bl=512, norm=True, clip=True, gains=(rgb.numpy(), red_g, blue_g), post_clip=True)
    assert np.array_equal(out.cpu().numpy(), g['gain_crops_clip'])


def test_dark_shading_in_front_of_pack(G):
    from pnnp_amd.augment import CropAugment
    g, meta = G
    H, W = g['frame'].shape
    ca = CropAugment({'H': H, 'W': W, 'patch_size': 64, 'crop_per_image': 1}, ways=4)
    ca.h_start, ca.w_start, ca.aug = [9], [21], np.array([0])
    frame = torch.from_numpy(g['frame']).cuda()
    sl = (slice(None), slice(9, 73), slice(21, 85))
    out = ca.crop_pack(frame, wp=16383, bl=512, norm=True, clip=False, darkshading=g['dark'])
    assert np.array_equal(out[0].cpu().numpy(), g['dark_lr'][sl])
    out = ca.crop_pack(frame, wp=16383, bl=512, norm=True, clip=False, darkshading=g['dark'], dark_add=np.float32(meta['dark']['mean']))
    assert np.array_equal(out[0].cpu().numpy(), g['dark_lr_d'][sl])
    dark64 = g['dark'].astype(np.float64) * 1.000001
    out = ca.crop_pack(frame, wp=16383, bl=512, norm=True, clip=False, darkshading=dark64)
    assert np.array_equal(out[0].cpu().numpy(), g['dark64_lr'][sl])


def test_full_size_properties():
    """BASELINE-size case (Sony frame 2848x4256, 16 crops of 512): crop -> inverse augmentation == plain slice."""
    from pnnp_amd.augment import CropAugment
    from pnnp_amd.isp_ops import raw2bayer
    g = torch.Generator().manual_seed(0)
    frame = torch.randint(0, 16384, (2848, 4256), generator=g, dtype=torch.int32).to(torch.uint16).cuda()
    ca = CropAugment({'H': 2848, 'W': 4256, 'patch_size': 512, 'crop_per_image': 16}, ways=8)
    np.random.seed(0)
    ca.init_random_crop_point(mode='random')
    out = ca.crop_pack(frame, wp=16383, bl=512, norm=True, clip=True)
    hr = raw2bayer(frame, wp=16383, bl=512, norm=True, clip=True)
    for i in range(16):
        ref = hr[:, ca.h_start[i]:ca.h_end[i], ca.w_start[i]:ca.w_end[i]]
        rot, flip = int(ca.aug[i]) % 4, int(ca.aug[i]) // 4
        ref = torch.rot90(ref, rot, (-2, -1))
        if flip:
            ref = ref.flip(-1)
        assert torch.equal(out[i], ref), i


def test_empty_and_invalid():
    from pnnp_amd import _lib
    from pnnp_amd.augment import CropAugment
    frame = torch.zeros(64, 64, dtype=torch.uint16, device='cuda')
    ca = CropAugment({'H': 64, 'W': 64, 'patch_size': 16, 'crop_per_image': 0}, ways=8)
    assert ca.crop_pack(frame).shape == (0, 4, 16, 16)
    ca = CropAugment({'H': 64, 'W': 64, 'patch_size': 48, 'crop_per_image': 1}, ways=8)
    ca.h_start, ca.w_start, ca.aug = [0], [0], np.array([0])
    with pytest.raises(_lib.PnnpError):
        ca.crop_pack(frame)                     # 2*48 > 64
    with pytest.raises(_lib.PnnpError):
        ca.crop_pack(frame.float())
