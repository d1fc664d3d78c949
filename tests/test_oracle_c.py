"""Pin the C oracle (oracle/pnnp_oracle.c): pack/unpack bit-exact vs the reference's
goldens; Philox known-answer vectors; sampler distribution vs the reference's own draws."""
import json
import os

import numpy as np

from oracle import cbind, isp_np
from _noise_stats import check_against_reference, check_row_variance_analytic


def test_philox_known_answers():
    # Random123 kat_vectors, philox4x32-10
    assert [hex(x) for x in cbind.philox([0, 0, 0, 0], [0, 0])] == ['0x6627e8d5', '0xe169c58d', '0xbc57ac4c', '0x9b00dbd8']
    assert [hex(x) for x in cbind.philox([0xffffffff] * 4, [0xffffffff] * 2)] == ['0x408f276d', '0x41c83b0e', '0xa20bc7c6', '0x6d5451fd']
    assert [hex(x) for x in cbind.philox([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0])] == \
        ['0xd16cfe09', '0x94fdcceb', '0x5001e420', '0x24126ea1']


def test_c_pack_bit_exact(golden_dir):
    g = np.load(os.path.join(golden_dir, 'pack_small.npz'))
    meta = json.load(open(os.path.join(golden_dir, 'pack_meta.json')))
    for c in meta['cases']:
        t = c['tag']
        got = cbind.pack(g[t + '_raw'], g[t + '_bias'] + c['bl'], c['wp'], norm=c['norm'], clip=c['clip'])
        assert np.array_equal(got.view(np.uint32), g[t + '_packed'].view(np.uint32)), t
    for (H, W) in [(16, 24), (64, 64)]:
        for (wp, bl) in [(16383, 512), (1023, 64)]:
            assert np.array_equal(cbind.unpack(g[f'unpack_H{H}W{W}wp{wp}_in'], wp, bl), g[f'unpack_H{H}W{W}wp{wp}_out'])


def test_c_sampler_distribution_vs_reference(golden_dir):
    def sample(y, p, flags, seed, offset):
        return cbind.noise_sample(y, cbind.param_rows([p]), flags, seed=seed, offset=offset)
    n = check_against_reference(golden_dir, sample, cbind.noise_flags)
    assert n >= 150


def test_c_sampler_row_variance():
    def sample(y, p, flags, seed, offset):
        return cbind.noise_sample(y, cbind.param_rows([p] * y.shape[0]), flags, seed=seed, offset=offset)
    check_row_variance_analytic(sample, cbind.noise_flags)
