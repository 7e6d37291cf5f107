"""CPU restatement of the data side (TEST INFRASTRUCTURE ONLY).

dataset/CT_pancreas_ids.py:143-173 (IdPosPanCTDataset.__getitem__): np.load, HU clip to [-91, 250], (x - 86.9) / 39.4,
(D,H,W) -> (H,W,D), float32 / uint8; then the monai (0.7.0, absent here: PARITY UNPINNED) transforms restated from their published
algorithm: RandCropByPosNegLabeld(pos=0.7, neg=0.3) = generate_pos_neg_label_crop_centers + correct_crop_centers + SpatialCrop,
RandFlipd(prob=0.4, spatial_axis=(0, 1)).  The rotate / contrast / zoom augmentations are not restated.
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
