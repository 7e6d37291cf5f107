"""Data side of the training scripts on the device.

dataset/CT_pancreas_ids.py:143-173: `.npy` scan [D,H,W] -> HU clip [-91, 250] -> (x - 86.9) / 39.4 -> [H,W,D] float32, label uint8;
then RandCropByPosNegLabeld(pos=0.7, neg=0.3, num_samples) and RandFlipd(prob=0.4, spatial_axis=(0, 1)) of monai 0.7.0.  A scan
is uploaded once; preprocessing, cropping and flipping are HIP kernels (csrc/data.hip).  Crop centres are drawn on the host from
the label's foreground / background index lists exactly as monai does (the label comes from disk, so it is host-resident anyway).
The rotate / contrast / zoom augmentations of the reference are not implemented yet.
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
