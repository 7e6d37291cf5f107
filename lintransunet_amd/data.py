"""Data side of the training scripts on the device.

dataset/CT_pancreas_ids.py:143-173: `.npy` scan [D,H,W] -> HU clip [-91, 250] -> (x - 86.9) / 39.4 -> [H,W,D] float32, label uint8;
then RandCropByPosNegLabeld(pos=0.7, neg=0.3, num_samples) and RandFlipd(prob=0.4, spatial_axis=(0, 1)) of monai 0.7.0.  A scan
is uploaded once; preprocessing, cropping and flipping are HIP kernels (csrc/data.hip).  Crop centres are drawn on the host from
the label's foreground / background index lists exactly as monai does (the label comes from disk, so it is host-resident anyway).
`augment` applies the remaining augmentations of the reference (RandRotated, RandAdjustContrastd, RandZoomd, RandFlipd;
CT_pancreas_ids.py:121-134) to a batch of patches on the device; the random draws stay on the host.
"""
import numpy as np
import torch

from . import _lib
from .ops import _p, _s

LOW_CLIP, HIGH_CLIP, MEAN, STD = -91.0, 250.0, 86.9, 39.4


def preprocess(raw_img, raw_label, device='cuda'):
    """raw [D,H,W] numpy arrays / tensors -> (img f32 [H,W,D], label u8 [H,W,D]) on the device"""
    ri = torch.as_tensor(np.ascontiguousarray(raw_img), dtype=torch.float32).to(device)
    rl = torch.as_tensor(np.ascontiguousarray(raw_label)).to(torch.uint8).to(device)
    if not ri.is_cuda:
        raise _lib.LtuError('data.preprocess runs on the GPU only (no CPU fallback)')
    D, H, W = ri.shape
    img = torch.empty((H, W, D), device=ri.device, dtype=torch.float32)
    lab = torch.empty((H, W, D), device=ri.device, dtype=torch.uint8)
    _lib.call('ltu_ct_preprocess', _p(ri), _p(img), _p(rl), _p(lab), D, H, W, LOW_CLIP, HIGH_CLIP, MEAN, STD, _s())
    return img, lab


def correct_crop_centers(centers, spatial_size, label_shape):
    """monai/transforms/utils.py::correct_crop_centers (0.7.0)"""
    out = []
    for c, s, n in zip(centers, spatial_size, label_shape):
        if n < s:
            raise ValueError('The size of the proposed random crop ROI is larger than the image size.')
        lo = s // 2
        hi = int(np.uint16(n + 1 - s / 2))
        if lo == hi:
            hi += 1
        out.append(int(min(max(c, lo), hi - 1)))
    return out


def crop_centers(label, spatial_size, num_samples, pos=0.7, neg=0.3, rand_state=None):
    """centres of RandCropByPosNegLabeld; label: host array [H,W,D]"""
    rs = rand_state or np.random.RandomState()
    flat = (np.asarray(label) > 0).ravel()
    fg, bg = np.nonzero(flat)[0], np.nonzero(~flat)[0]
    pos_ratio = pos / (pos + neg)
    if fg.size == 0 and bg.size == 0:
        raise ValueError('No sampling location available.')
    if fg.size == 0 or bg.size == 0:
        pos_ratio = 0 if fg.size == 0 else 1
    centers = []
    for _ in range(num_samples):
        use = fg if rs.rand() < pos_ratio else bg
        idx = use[rs.randint(len(use))]
        centers.append(correct_crop_centers(list(np.unravel_index(idx, np.asarray(label).shape)), spatial_size, np.asarray(label).shape))
    return centers


def crop_flip(img, lab, centers, flips, spatial_size):
    """device patches: ([n,1,h,w,d] f32, [n,1,h,w,d] u8) from img / lab [H,W,D] at the given centres; flips[k] mirrors H and W"""
    H, W, D = img.shape
    h, w, d = spatial_size
    desc = torch.tensor([[max(c[0] - h // 2, 0), max(c[1] - w // 2, 0), max(c[2] - d // 2, 0), int(f), int(f)] for c, f in zip(centers, flips)],
                        dtype=torch.int32).to(img.device)
    n = len(centers)
    oi = torch.empty((n, 1, h, w, d), device=img.device, dtype=torch.float32)
    ol = torch.empty((n, 1, h, w, d), device=img.device, dtype=torch.uint8)
    _lib.call('ltu_crop_flip', _p(img), _p(oi), _p(desc), n, H, W, D, h, w, d, 4, _s())
    _lib.call('ltu_crop_flip', _p(lab), _p(ol), _p(desc), n, H, W, D, h, w, d, 1, _s())
    return oi, ol


def sample_patches(img, lab, label_host, spatial_size, num_samples, rand_state, flip_prob=0.4):
    """one `__getitem__` of IdPosPanCTDataset without the rotate / contrast / zoom augmentations"""
    centers = crop_centers(label_host, spatial_size, num_samples, rand_state=rand_state)
    flips = [rand_state.rand() < flip_prob for _ in range(num_samples)]
    return crop_flip(img, lab, centers, flips, spatial_size)


# ---- augmentations (CT_pancreas_ids.py:121-134) -----------------------------------------------------------------------------

def rotate_matrix(angles, shape):
    """monai create_rotate (Rx @ Ry @ Rz) about the patch centre, as a 3x4 float32 pull matrix (output voxel -> input location)"""
    ax, ay, az = (float(a) for a in angles)
    rx = np.array([[1, 0, 0, 0], [0, np.cos(ax), -np.sin(ax), 0], [0, np.sin(ax), np.cos(ax), 0], [0, 0, 0, 1]], dtype=np.float64)
    ry = np.array([[np.cos(ay), 0, np.sin(ay), 0], [0, 1, 0, 0], [-np.sin(ay), 0, np.cos(ay), 0], [0, 0, 0, 1]], dtype=np.float64)
    rz = np.array([[np.cos(az), -np.sin(az), 0, 0], [np.sin(az), np.cos(az), 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]], dtype=np.float64)
    c = (np.asarray(shape, dtype=np.float64) - 1) / 2
    sh, sh1 = np.eye(4), np.eye(4)
    sh[:3, 3], sh1[:3, 3] = c, -c
    return (sh @ rx @ ry @ rz @ sh1)[:3].astype(np.float32)


_IDENTITY = np.eye(4, dtype=np.float32)[:3]


def draw_augmentation(rs, rot_prob=0.1, rot_range=np.pi / 9, prob=0.4, zoom=(0.7, 1.3), gamma=(0.5, 4.5)):
    """one sample's random parameters in the reference's transform order; every draw is made whether or not the transform fires"""
    p = {}
    p['rotate'] = rs.rand() < rot_prob
    p['angles'] = [rs.uniform(-rot_range, rot_range) for _ in range(3)]
    p['contrast'] = rs.rand() < prob
    p['gamma'] = rs.uniform(*gamma)
    p['zoom'] = rs.rand() < prob
    p['zoom_factor'] = rs.uniform(*zoom)
    p['flip'] = rs.rand() < prob
    return p


def _vol4(t):
    if not t.is_cuda:
        raise _lib.LtuError('data augmentations run on the GPU only (no CPU fallback)')
    n = t.shape[0]
    return t.reshape(n, *t.shape[-3:]).contiguous()


def rotate(x, mats):
    """x [n,(1,)H,W,D] f32, mats [n,3,4]: trilinear pull resampling with border padding (monai Rotate, keep_size)"""
    v = _vol4(x)
    n, H, W, D = v.shape
    m = torch.as_tensor(np.ascontiguousarray(mats, dtype=np.float32).reshape(n, 12)).to(v.device)
    out = torch.empty_like(v)
    _lib.call('ltu_affine_sample', _p(v), _p(out), _p(m), n, H, W, D, _s())
    return out.view(x.shape)


def zoom(x, factors):
    """monai Zoom(keep_size=True): trilinear align_corners interpolation to floor(size*f) + centred edge pad / crop"""
    v = _vol4(x)
    n, H, W, D = v.shape
    # output size of F.interpolate(scale_factor=f): floor(size * f) in double precision
    z = torch.tensor([[max(int(np.floor(float(sz) * float(f))), 1) for sz in (H, W, D)] for f in factors], dtype=torch.int32).to(v.device)
    out = torch.empty_like(v)
    _lib.call('ltu_zoom_sample', _p(v), _p(out), _p(z), n, H, W, D, _s())
    return out.view(x.shape)


def adjust_contrast(x, gammas):
    """monai AdjustContrast per patch; gamma <= 0 leaves a patch untouched"""
    v = _vol4(x)
    n = v.shape[0]
    g = torch.as_tensor(np.asarray(gammas, dtype=np.float32)).to(v.device)
    ws = torch.empty(2 * n, device=v.device, dtype=torch.int32)
    out = torch.empty_like(v)
    _lib.call('ltu_adjust_contrast', _p(v), _p(out), _p(g), _p(ws), n, v.numel() // n, _s())
    return out.view(x.shape)


def augment(img, lab, params):
    """img f32 [n,1,h,w,d], lab u8 [n,1,h,w,d], params: one draw_augmentation() dict per patch -> augmented (f32, u8).
    The label is resampled in float and truncated back to uint8 as the reference does (CT_pancreas_ids.py:171)."""
    n = img.shape[0]
    shape = tuple(img.shape[-3:])
    lf = lab.to(torch.float32)
    if any(p['rotate'] for p in params):
        mats = np.stack([rotate_matrix(p['angles'], shape) if p['rotate'] else _IDENTITY for p in params])
        img, lf = rotate(img, mats), rotate(lf, mats)
    if any(p['contrast'] for p in params):
        img = adjust_contrast(img, [p['gamma'] if p['contrast'] else -1.0 for p in params])
    if any(p['zoom'] for p in params):
        f = [p['zoom_factor'] if p['zoom'] else 1.0 for p in params]
        img, lf = zoom(img, f), zoom(lf, f)
    lab8 = lf.to(torch.uint8)
    if any(p['flip'] for p in params):
        h, w, d = shape
        flips = [p['flip'] for p in params]
        oi = torch.empty_like(img)
        ol = torch.empty_like(lab8)
        desc = torch.tensor([[0, 0, 0, int(f), int(f)] for f in flips], dtype=torch.int32).to(img.device)
        for k in range(n):                       # every patch is its own "volume" here: crop of the full extent, optional mirror
            _lib.call('ltu_crop_flip', _p(img[k]), _p(oi[k]), _p(desc[k]), 1, h, w, d, h, w, d, 4, _s())
            _lib.call('ltu_crop_flip', _p(lab8[k]), _p(ol[k]), _p(desc[k]), 1, h, w, d, h, w, d, 1, _s())
        img, lab8 = oi, ol
    return img, lab8


def synthetic_patches(batch, size, seed, device, n_classes=2, n_blobs=2):
    """Synthetic CT-like training patches for benchmarks and soak runs: N(0,1) intensities clipped to
    the dataset's normalised HU range [(-91 - 86.9) / 39.4, (250 - 86.9) / 39.4] and a label volume that is a union of `n_blobs`
    random ellipsoids per patch (nested twice for 3 label values).  Host generators (seeded), uploaded once: a benchmark keeps its
    inputs resident in HBM.  Returns (x f32 [B,1,H,W,D], label u8 [B,1,H,W,D])."""
    g = torch.Generator().manual_seed(seed)
    H, W, D = size
    x = torch.randn((batch, 1, H, W, D), generator=g).clamp_((LOW_CLIP - MEAN) / STD, (HIGH_CLIP - MEAN) / STD)
    hh = torch.arange(H, dtype=torch.float32).view(H, 1, 1)
    ww = torch.arange(W, dtype=torch.float32).view(1, W, 1)
    dd = torch.arange(D, dtype=torch.float32).view(1, 1, D)
    lab = torch.zeros((batch, 1, H, W, D), dtype=torch.uint8)
    for b in range(batch):
        for _ in range(n_blobs):
            c = 0.25 + 0.5 * torch.rand(3, generator=g)
            r = 0.12 + 0.15 * torch.rand(3, generator=g)
            dist = ((hh - c[0] * H) / (r[0] * H)) ** 2 + ((ww - c[1] * W) / (r[1] * W)) ** 2 + ((dd - c[2] * D) / (r[2] * D)) ** 2
            lab[b, 0][dist <= 1.0] = 1
            if n_classes == 3:
                lab[b, 0][dist <= 0.25] = 2
    return x.to(device), lab.to(device)
