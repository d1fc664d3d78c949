"""Running a persistent convolution grid NEXT to another resident kernel (gradient all-reduce
overlapped with the backward pass, base_trainer.py:115-118).  The 1-GPU boxes cannot run RCCL with two ranks, so a kernel that
occupies k CUs stands in for the collective (tools/ubench/squat.hip)."""
import os
import shutil
import sys

import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))

pytestmark = pytest.mark.gpu


def test_persistent_split_bounds_the_cost_of_occupied_cus():
    """With one workgroup per CU and static shares, the workgroups whose CUs are occupied start when the others have FINISHED and the
    layer takes about twice as long.  With pnnp_set_persistent_split(4) the hardware dispatcher hands the quarter shares to whichever
    CU frees up.  conv2_2 forward has 2048 tiles (8 per CU); at k = 32 the 1024 quarter shares of 2 tiles need ceil(1024 / 224) = 5
    rounds = 10 tile times against 8 alone: x 1.25 is what the tile granularity allows any dynamic scheme (+ 12 % for the smaller
    shares' pipeline fills and box noise; measured x 1.04-1.05: the chip is power-limited, 224 CUs clock higher than 256); the result is
    bit-identical, and alone on the chip the split costs < 20 % (measured 6-14 %), which is why it is not the default."""
    if not (shutil.which('hipcc') or os.path.exists('/opt/rocm/bin/hipcc')):
        pytest.skip('hipcc not available to build the occupying kernel')
    import squat_test
    k = 32
    r = squat_test.measure(k)
    print(r)
    assert r['same_result']
    assert r['alone_4'] < 1.20 * r['alone_1'], r                       # (measured 6-14 %: four pipeline fills per CU instead of one)
    assert r['beside_1'] > 1.3 * r['alone_1'], r                       # the hazard exists (or this test proves nothing); measured x 1.45
    assert r['beside_4'] < r['beside_1'] * 0.85, r
    assert r['beside_4'] < 1.25 * 1.12 * r['alone_1'], r
