"""CPU: the host half of rows f2 (crop points, augmentation modes, WB gains) reproduces the reference's
RNG draws (tests/golden/augment.json, made by the reference's dataset classes)."""
import json
import os

import numpy as np
import pytest
import torch


@pytest.mark.parametrize('tag,ways', [('syn_random', 8), ('syn_grid', 8), ('real_random', 4)])
def test_crop_points(golden_dir, tag, ways):
    from pnnp_amd.augment import CropAugment
    m = json.load(open(os.path.join(golden_dir, 'augment.json')))[tag]
    ca = CropAugment({'H': 160, 'W': 224, 'patch_size': m['ps'], 'crop_per_image': m['crop_per_image']}, ways=ways)
    np.random.seed(m['seed'])
    ca.init_random_crop_point(mode=m['mode'])
    assert [int(v) for v in ca.h_start] == m['h_start']
    assert [int(v) for v in ca.w_start] == m['w_start']
    assert [int(v) for v in ca.aug] == m['aug']
    assert all(e - s == m['ps'] for s, e in zip(ca.h_start, ca.h_end))


def test_random_gains(golden_dir):
    from pnnp_amd.augment import random_gains
    mg = json.load(open(os.path.join(golden_dir, 'augment.json')))['gains']
    torch.manual_seed(mg['torch_seed']); np.random.seed(mg['np_seed'])
    rgb, red, blue = random_gains()
    assert (float(rgb[0]), float(red[0]), float(blue[0])) == (mg['rgb'], mg['red_raw'], mg['blue_raw'])
    with pytest.raises(NotImplementedError):
        random_gains('NikonD850')


def test_no_cpu_path():
    from pnnp_amd import _lib
    from pnnp_amd.augment import CropAugment
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    ca = CropAugment({'H': 64, 'W': 64, 'patch_size': 16, 'crop_per_image': 1})
    ca.h_start, ca.w_start = [0], [0]
    with pytest.raises((_lib.PnnpError, RuntimeError, AssertionError)):
        ca.random_crop(torch.zeros(4, 32, 32))
