"""Data side: preprocessing, biased crop centres, crop + flip."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import data as O  # noqa: E402


def _scan(seed, shape=(21, 40, 36)):
    rs = np.random.RandomState(seed)
    raw = (rs.randn(*shape) * 180 + 40).astype(np.float32)
    lab = np.zeros(shape, np.uint8)
    lab[6:14, 10:25, 8:20] = 1
    return raw, lab


def test_oracle_preprocess_and_centres():
    raw, lab = _scan(0)
    img, l2 = O.preprocess(raw, lab)
    assert img.shape == (40, 36, 21) and l2.shape == (40, 36, 21) and img.dtype == np.float32
    assert abs(img.max() - (250 - 86.9) / 39.4) < 1e-5 and abs(img.min() - (-91 - 86.9) / 39.4) < 1e-5
    rs = np.random.RandomState(3)
    ctrs = O.crop_centers(l2, (16, 16, 8), 200, rand_state=rs)
    starts = O.crop_starts(ctrs, (16, 16, 8))
    assert all(0 <= s[i] and s[i] + (16, 16, 8)[i] <= l2.shape[i] for s in starts for i in range(3))
    on_fg = np.mean([l2[c[0], c[1], c[2]] for c in ctrs])          # ~0.7 of the draws start from a foreground voxel
    assert 0.4 < on_fg <= 1.0


def test_host_centres_match_oracle():
    from lintransunet_amd import data as P
    _, lab = _scan(1)
    l2 = lab.transpose(1, 2, 0)
    a = O.crop_centers(l2, (16, 12, 8), 50, rand_state=np.random.RandomState(9))
    b = P.crop_centers(l2, (16, 12, 8), 50, rand_state=np.random.RandomState(9))
    assert a == b
    for shape, size in [((40, 36, 21), (40, 36, 21)), ((17, 9, 8), (8, 8, 8)), ((33, 33, 32), (32, 32, 32))]:
        for c in [(0, 0, 0), tuple(s - 1 for s in shape), tuple(s // 2 for s in shape)]:
            assert O.correct_crop_centers(list(c), size, shape) == P.correct_crop_centers(list(c), size, shape)


@pytest.mark.gpu
def test_device_pipeline_matches_oracle():
    from lintransunet_amd import data as P
    raw, lab = _scan(2)
    img, l2 = P.preprocess(raw, lab)
    oi, ol = O.preprocess(raw, lab)
    assert np.array_equal(img.cpu().numpy(), oi) and np.array_equal(l2.cpu().numpy(), ol)      # same fp32 operations: bit-exact
    size = (16, 12, 8)
    ctrs = O.crop_centers(ol, size, 6, rand_state=np.random.RandomState(4))
    flips = [True, False, True, False, False, True]
    pi, pl = P.crop_flip(img, l2, ctrs, flips, size)
    starts = O.crop_starts(ctrs, size)
    assert np.array_equal(pi[:, 0].cpu().numpy(), O.crop_flip(oi, starts, flips, size))
    assert np.array_equal(pl[:, 0].cpu().numpy(), O.crop_flip(ol, starts, flips, size))
    bi, bl = P.sample_patches(img, l2, ol, size, 3, np.random.RandomState(5))
    assert bi.shape == (3, 1, 16, 12, 8) and bl.dtype == torch.uint8
