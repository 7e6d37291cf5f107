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


# ---- augmentations (rotate / contrast / zoom) ---------------------------------------------------------------------------------

def _patch(seed, shape=(20, 17, 13)):
    rs = np.random.RandomState(seed)
    v = rs.randn(*shape).astype(np.float32)
    lab = np.zeros(shape, np.uint8)
    lab[5:14, 4:12, 3:9] = 1
    return v, lab


def test_oracle_rotate_is_grid_sample():
    """the oracle's pull resampling against torch's affine_grid + grid_sample (what monai's AffineTransform calls): bilinear,
    border padding, align_corners=True.  torch goes through normalised coordinates, hence the tolerance."""
    import torch.nn.functional as F
    v, _ = _patch(0)
    H, W, D = v.shape
    m = O.rotate_matrix((0.21, -0.3, 0.17), v.shape)
    out = O.affine_sample(v, m)
    # voxel-space pull matrix -> normalised theta in torch's (x = last dim, y, z = first dim) order
    size = np.array([H, W, D], dtype=np.float64)
    A = np.eye(4); A[:3] = m.astype(np.float64)
    N = np.eye(4); N[:3, :3] = np.diag((size - 1) / 2); N[:3, 3] = (size - 1) / 2          # normalised -> voxel
    T = np.linalg.inv(N) @ A @ N                                                             # normalised out -> normalised in
    P = np.eye(4)[[2, 1, 0, 3]]                                                              # (i,j,k) <-> (x,y,z)
    theta = torch.from_numpy((P @ T @ P)[:3]).float()[None]
    grid = F.affine_grid(theta, (1, 1, H, W, D), align_corners=True)
    ref = F.grid_sample(torch.from_numpy(v)[None, None], grid, mode='bilinear', padding_mode='border', align_corners=True)[0, 0].numpy()
    assert np.abs(out - ref).max() < 2e-4
    assert np.array_equal(O.affine_sample(v, np.eye(4, dtype=np.float32)[:3]), v)          # identity is exact


@pytest.mark.parametrize('zoom', [0.7, 0.93, 1.0, 1.18, 1.3])
def test_oracle_zoom_is_interpolate(zoom):
    """the fused oracle against F.interpolate(scale_factor, trilinear, align_corners=True) + monai's centred edge pad / crop"""
    import torch.nn.functional as F
    v, _ = _patch(1)
    out = O.zoom_sample(v, zoom)
    z = F.interpolate(torch.from_numpy(v)[None, None], scale_factor=[float(zoom)] * 3, mode='trilinear', align_corners=True)[0, 0].numpy()
    pads, sl = [], []
    for n, zn in zip(v.shape, z.shape):
        diff = n - zn
        half = abs(diff) // 2
        pads.append((half, diff - half) if diff > 0 else (0, 0))
        sl.append(slice(half, half + n) if diff < 0 else slice(None))
    ref = np.pad(z, pads, mode='edge')[tuple(sl)]
    assert ref.shape == v.shape
    assert np.abs(out - ref).max() < 1e-5
    if zoom == 1.0:
        assert np.array_equal(out, v)


def test_oracle_contrast_and_draws():
    v, lab = _patch(2)
    assert np.abs(O.adjust_contrast(v, 1.0) - v).max() < 1e-5
    c = O.adjust_contrast(v, 2.5)
    assert abs(c.min() - v.min()) < 1e-5 and abs(c.max() - v.max()) < 1e-4 and (c <= v + 1e-5).all()
    from lintransunet_amd import data as P
    a, b = O.draw_augmentation(np.random.RandomState(11)), P.draw_augmentation(np.random.RandomState(11))
    assert a == b
    assert np.array_equal(O.rotate_matrix(a['angles'], v.shape), P.rotate_matrix(a['angles'], v.shape))
    ps = [O.draw_augmentation(np.random.RandomState(s)) for s in range(400)]
    assert 0.05 < np.mean([p['rotate'] for p in ps]) < 0.16 and 0.32 < np.mean([p['zoom'] for p in ps]) < 0.48
    assert all(0.7 <= p['zoom_factor'] <= 1.3 and 0.5 <= p['gamma'] <= 4.5 and max(map(abs, p['angles'])) <= np.pi / 9 for p in ps)
    oi, ol = O.augment(v, lab, dict(rotate=False, contrast=False, zoom=False, flip=False, angles=[0, 0, 0], gamma=1, zoom_factor=1))
    assert np.array_equal(oi, v) and np.array_equal(ol, lab)


@pytest.mark.gpu
def test_device_augmentations_match_oracle():
    from lintransunet_amd import data as P
    vols = [_patch(s) for s in (3, 4, 5, 6)]
    img = torch.from_numpy(np.stack([v for v, _ in vols]))[:, None].cuda()
    lab = torch.from_numpy(np.stack([l for _, l in vols]))[:, None].cuda()
    shape = vols[0][0].shape
    # each transform alone, per-sample parameters (identity / zoom 1 / gamma <= 0 leave a sample untouched, exactly)
    angles = [(0.3, -0.2, 0.1), (0, 0, 0), (-0.34, 0.34, 0.0), (0.05, 0.0, -0.3)]
    mats = np.stack([P.rotate_matrix(a, shape) for a in angles])
    r = P.rotate(img, mats).cpu().numpy()
    for k, (v, _) in enumerate(vols):
        assert np.abs(r[k, 0] - O.affine_sample(v, mats[k])).max() < 2e-5
    assert np.array_equal(r[1, 0], vols[1][0])
    zf = [0.7, 1.0, 1.3, 0.88]
    z = P.zoom(img, zf).cpu().numpy()
    for k, (v, _) in enumerate(vols):
        assert np.abs(z[k, 0] - O.zoom_sample(v, zf[k])).max() < 1e-5
    assert np.array_equal(z[1, 0], vols[1][0])
    g = [0.5, -1.0, 4.5, 2.0]
    c = P.adjust_contrast(img, g).cpu().numpy()
    for k, (v, _) in enumerate(vols):
        ref = v if g[k] <= 0 else O.adjust_contrast(v, g[k])
        assert np.abs(c[k, 0] - ref).max() < 2e-5 * max(1.0, np.abs(ref).max())
    # the whole chain with drawn parameters
    params = [P.draw_augmentation(np.random.RandomState(s)) for s in (21, 22, 23, 24)]
    params[0].update(rotate=True, zoom=True, contrast=True, flip=True)
    params[1].update(rotate=False, zoom=False, contrast=False, flip=False)
    ai, al = P.augment(img, lab, params)
    assert ai.dtype == torch.float32 and al.dtype == torch.uint8
    for k, (v, l) in enumerate(vols):
        oi, ol = O.augment(v, l, params[k])
        assert np.abs(ai[k, 0].cpu().numpy() - oi).max() < 1e-4 * max(1.0, np.abs(oi).max())
        assert (al[k, 0].cpu().numpy() != ol).mean() < 2e-3          # a label voxel interpolated to 1 - 1e-7 may truncate differently
    assert np.array_equal(ai[1, 0].cpu().numpy(), vols[1][0]) and np.array_equal(al[1, 0].cpu().numpy(), vols[1][1])


@pytest.mark.gpu
def test_device_augmentations_full_size_properties():
    """at the reference's patch size (512 x 512 x 32): identities are exact, a rotation followed by its inverse returns the
    interior, zoom keeps the extrema inside the input range, contrast keeps min / max"""
    from lintransunet_amd import data as P
    g = torch.Generator().manual_seed(5)
    img = torch.randn(2, 1, 512, 512, 32, generator=g).cuda()
    shape = (512, 512, 32)
    ident = np.stack([np.eye(4, dtype=np.float32)[:3]] * 2)
    assert torch.equal(P.rotate(img, ident), img)
    assert torch.equal(P.zoom(img, [1.0, 1.0]), img)
    assert torch.equal(P.adjust_contrast(img, [-1.0, 0.0]), img)
    ii, jj, kk = torch.meshgrid(torch.arange(512.), torch.arange(512.), torch.arange(32.), indexing='ij')
    sm = (torch.sin(ii / 40) + torch.cos(jj / 30) + torch.sin(kk / 9))[None, None].repeat(2, 1, 1, 1, 1).cuda()   # smooth field
    m = P.rotate_matrix((0.0, 0.0, 0.2), shape)
    A = np.eye(4); A[:3] = m
    back = np.linalg.inv(A)[:3].astype(np.float32)
    rr = P.rotate(P.rotate(sm, np.stack([m, m])), np.stack([back, back]))
    inner = (slice(None), slice(None), slice(160, 352), slice(160, 352), slice(None))
    assert (rr[inner] - sm[inner]).abs().max().item() < 2e-3
    z = P.zoom(img, [0.7, 1.3])
    assert z.min() >= img.min() and z.max() <= img.max()
    c = P.adjust_contrast(img, [0.5, 4.5])
    for k in range(2):
        assert abs(c[k].min().item() - img[k].min().item()) < 1e-4 and abs(c[k].max().item() - img[k].max().item()) < 1e-3
