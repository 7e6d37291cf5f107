"""CPU restatement of the data side (TEST INFRASTRUCTURE ONLY).

dataset/CT_pancreas_ids.py:143-173 (IdPosPanCTDataset.__getitem__): np.load, HU clip to [-91, 250], (x - 86.9) / 39.4,
(D,H,W) -> (H,W,D), float32 / uint8; then the monai (0.7.0, absent here: PARITY UNPINNED) transforms restated from their published
algorithm: RandCropByPosNegLabeld(pos=0.7, neg=0.3) = generate_pos_neg_label_crop_centers + correct_crop_centers + SpatialCrop,
RandFlipd(prob=0.4, spatial_axis=(0, 1)), and (CT_pancreas_ids.py:121-133) RandRotated(range pi/9 per axis, prob 0.1 = monai's
default, bilinear, border padding, align_corners), RandAdjustContrastd(prob 0.4, gamma in (0.5, 4.5)), RandZoomd(prob 0.4, zoom in
(0.7, 1.3), trilinear, align_corners, keep_size with edge padding).  The resampling arithmetic is torch's grid_sample /
upsample_trilinear3d, which monai calls and which tests/test_data.py pins on the CPU; monai's matrix conventions are from memory.
"""
import numpy as np

LOW_CLIP, HIGH_CLIP, MEAN, STD = -91, 250, 86.9, 39.4


def preprocess(raw_img, raw_label):
    """CT_pancreas_ids.py:150-158 on float32 input"""
    img = raw_img.astype(np.float32).copy()
    img[img < LOW_CLIP] = LOW_CLIP
    img[img > HIGH_CLIP] = HIGH_CLIP
    img = (img - np.float32(MEAN)) / np.float32(STD)
    return img.transpose((1, 2, 0)).astype(np.float32), raw_label.transpose((1, 2, 0)).astype(np.uint8)


def correct_crop_centers(centers, spatial_size, label_shape):
    """monai/transforms/utils.py::correct_crop_centers (0.7.0)"""
    spatial_size, label_shape = np.asarray(spatial_size), np.asarray(label_shape)
    if not (label_shape - spatial_size >= 0).all():
        raise ValueError('The size of the proposed random crop ROI is larger than the image size.')
    valid_start = np.floor_divide(spatial_size, 2)
    valid_end = np.subtract(label_shape + np.array(1), spatial_size / np.array(2)).astype(np.uint16)
    for i, vs in enumerate(valid_start):
        if vs == valid_end[i]:
            valid_end[i] += 1
    out = []
    for i, c in enumerate(centers):
        ci = c
        if c < valid_start[i]:
            ci = valid_start[i]
        if c >= valid_end[i]:
            ci = valid_end[i] - 1
        out.append(int(ci))
    return out


def crop_centers(label, spatial_size, num_samples, pos=0.7, neg=0.3, rand_state=None):
    """map_binary_to_indices + generate_pos_neg_label_crop_centers (0.7.0); label [H,W,D]"""
    rs = rand_state or np.random.RandomState()
    flat = (label > 0).ravel()
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
        centers.append(correct_crop_centers(list(np.unravel_index(idx, label.shape)), spatial_size, label.shape))
    return centers


def crop_starts(centers, spatial_size):
    """SpatialCrop(roi_center, roi_size): start = max(center - size // 2, 0)"""
    return [[max(c - s // 2, 0) for c, s in zip(ctr, spatial_size)] for ctr in centers]


def crop_flip(vol, starts, flips, spatial_size):
    """patches [n, h, w, d]; flips[k] mirrors axes 0 and 1 together (RandFlipd spatial_axis=(0, 1))"""
    out = []
    for st, fl in zip(starts, flips):
        p = vol[st[0]:st[0] + spatial_size[0], st[1]:st[1] + spatial_size[1], st[2]:st[2] + spatial_size[2]]
        if fl:
            p = np.flip(p, (0, 1))
        out.append(np.ascontiguousarray(p))
    return np.stack(out)


# ---- augmentations (monai 0.7.0 Rotate / AdjustContrast / Zoom restated) ---------------------------------------------------

def rotate_matrix(angles, shape):
    """monai create_rotate (3D: Rx @ Ry @ Rz) about the volume centre: shift((n-1)/2) @ R @ shift(-(n-1)/2); the matrix maps an
    OUTPUT voxel index to the INPUT location that is sampled (pull resampling, as AffineTransform / grid_sample do)."""
    ax, ay, az = (float(a) for a in angles)
    rx = np.array([[1, 0, 0, 0], [0, np.cos(ax), -np.sin(ax), 0], [0, np.sin(ax), np.cos(ax), 0], [0, 0, 0, 1]], dtype=np.float64)
    ry = np.array([[np.cos(ay), 0, np.sin(ay), 0], [0, 1, 0, 0], [-np.sin(ay), 0, np.cos(ay), 0], [0, 0, 0, 1]], dtype=np.float64)
    rz = np.array([[np.cos(az), -np.sin(az), 0, 0], [np.sin(az), np.cos(az), 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]], dtype=np.float64)
    c = (np.asarray(shape, dtype=np.float64) - 1) / 2
    sh, sh1 = np.eye(4), np.eye(4)
    sh[:3, 3], sh1[:3, 3] = c, -c
    return (sh @ rx @ ry @ rz @ sh1)[:3].astype(np.float32)


def affine_sample(vol, mat):
    """out[p] = trilinear(vol, mat @ (p, 1)) with border padding (coordinate clamped to [0, n-1]); vol [H,W,D] f32, mat [3,4] f32"""
    H, W, D = vol.shape
    m = mat.astype(np.float32)
    i, j, k = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), np.arange(D, dtype=np.float32), indexing='ij')
    co = []
    for r, n in zip(range(3), (H, W, D)):
        base = m[r, 0] * i + m[r, 3]                       # same association as the kernel: (m0*i + m3) + m1*j + m2*k
        c = base + m[r, 1] * j + m[r, 2] * k
        co.append(np.clip(c.astype(np.float32), np.float32(0), np.float32(n - 1)))
    f = [np.floor(c) for c in co]
    t = [c - fl for c, fl in zip(co, f)]
    i0 = [fl.astype(np.int64) for fl in f]
    out = np.zeros(vol.shape, dtype=np.float32)
    for a in range(2):
        for b in range(2):
            for c in range(2):
                xi, yi, zi = i0[0] + a, i0[1] + b, i0[2] + c
                w = ((t[0] if a else 1 - t[0]) * (t[1] if b else 1 - t[1]) * (t[2] if c else 1 - t[2])).astype(np.float32)
                ok = (xi < H) & (yi < W) & (zi < D)
                out += np.where(ok, w * vol[np.minimum(xi, H - 1), np.minimum(yi, W - 1), np.minimum(zi, D - 1)], np.float32(0)).astype(np.float32)
    return out


def _zoom_axis(size, zoom):
    Z = max(int(np.floor(float(size) * float(zoom))), 1)          # torch: floor(double(size) * scale_factor)
    diff = size - Z
    half = abs(diff) // 2
    q = np.clip(np.arange(size) + (-half if diff > 0 else half), 0, Z - 1)
    scale = np.float32(size - 1) / np.float32(Z - 1) if Z > 1 else np.float32(0)
    src = (scale * q.astype(np.float32)).astype(np.float32)
    x0 = np.minimum(src.astype(np.int64), size - 1)
    x1 = x0 + (x0 < size - 1)
    return x0, x1, (src - x0.astype(np.float32)).astype(np.float32)


def zoom_sample(vol, zoom):
    """monai Zoom(keep_size=True, padding_mode='edge'): interpolate(scale_factor=zoom, trilinear, align_corners=True) to
    floor(n * zoom), then centred edge pad / centre crop back; vol [H,W,D] f32"""
    H, W, D = vol.shape
    (h0, h1, lh), (w0, w1, lw), (d0, d1, ld) = _zoom_axis(H, zoom), _zoom_axis(W, zoom), _zoom_axis(D, zoom)
    one = np.float32(1)
    def along_d(hh, ww):
        r = vol[hh][:, ww]                                   # [H, W, D] rows gathered
        return (one - ld) * r[:, :, d0] + ld * r[:, :, d1]
    c00, c01, c10, c11 = along_d(h0, w0), along_d(h0, w1), along_d(h1, w0), along_d(h1, w1)
    lw_, lh_ = lw[None, :, None], lh[:, None, None]
    c0 = (one - lw_) * c00 + lw_ * c01
    c1 = (one - lw_) * c10 + lw_ * c11
    return ((one - lh_) * c0 + lh_ * c1).astype(np.float32)


def adjust_contrast(vol, gamma):
    """monai AdjustContrast: ((x - min) / (range + 1e-7)) ** gamma * range + min, float32"""
    lo = vol.min()
    rng = vol.max() - lo
    return (np.power((vol - lo) / np.float32(rng + np.float32(1e-7)), np.float32(gamma)) * rng + lo).astype(np.float32)


def draw_augmentation(rs, rot_prob=0.1, rot_range=np.pi / 9, prob=0.4, zoom=(0.7, 1.3), gamma=(0.5, 4.5)):
    """one sample's random parameters, drawn in the reference's transform order (rotate, contrast, zoom, flip); every draw is
    made whether or not the transform fires, as monai's randomize() methods do"""
    p = {}
    p['rotate'] = rs.rand() < rot_prob
    p['angles'] = [rs.uniform(-rot_range, rot_range) for _ in range(3)]
    p['contrast'] = rs.rand() < prob
    p['gamma'] = rs.uniform(*gamma)
    p['zoom'] = rs.rand() < prob
    p['zoom_factor'] = rs.uniform(*zoom)
    p['flip'] = rs.rand() < prob
    return p


def augment(img, lab, params):
    """img f32 [H,W,D], lab u8 [H,W,D] -> augmented (f32, u8); the label is resampled in float and truncated back, as the
    reference's `.to(torch.uint8)` does (CT_pancreas_ids.py:171)"""
    lf = lab.astype(np.float32)
    if params['rotate']:
        m = rotate_matrix(params['angles'], img.shape)
        img, lf = affine_sample(img, m), affine_sample(lf, m)
    if params['contrast']:
        img = adjust_contrast(img, params['gamma'])
    if params['zoom']:
        img, lf = zoom_sample(img, params['zoom_factor']), zoom_sample(lf, params['zoom_factor'])
    if params['flip']:
        img, lf = np.flip(img, (0, 1)), np.flip(lf, (0, 1))
    return np.ascontiguousarray(img), np.ascontiguousarray(lf).astype(np.uint8)
