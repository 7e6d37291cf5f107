"""CPU restatement of the sliding-window inference driver (TEST INFRASTRUCTURE ONLY).

Reference call site: inference_embed_attn.py:141
    predict = sliding_window_inference(images, (512, 512, depth), 4, model, overlap=0.6, sigma_scale=0)
    predict2 = (predict >= 0.5).float();  metrics = [DiceClassLoss, Recall, Precision, LocalizationLoss](predict2, masks)

`sliding_window_inference` is monai's (requirements.txt: monai==0.7.0), which is absent from /root/reference and from this
image: PARITY UNPINNED for the window scheduling / blending.  This file restates the published algorithm of
monai/inferers/utils.py::sliding_window_inference (0.7.0), monai/data/utils.py::dense_patch_slices and
monai/inferers/utils.py::_get_scan_interval for mode="constant" (importance map = 1), padding_mode="constant", cval=0.
The metrics restate loss/criterions.py (Recall 280-311, Precision 348-379, LocalizationLoss 179-241, DiceClassLoss 35-70) and
ARE pinned by tests/golden/metrics.npz, generated from the reference's own modules.
"""
import math

import torch
import torch.nn.functional as F


def scan_interval(image_size, roi_size, overlap):
    """_get_scan_interval: roi == image -> roi, else int(roi * (1 - overlap)), at least 1"""
    out = []
    for img, roi in zip(image_size, roi_size):
        if roi == img:
            out.append(int(roi))
        else:
            iv = int(roi * (1 - overlap))
            out.append(iv if iv > 0 else 1)
    return tuple(out)


def patch_starts(image_size, roi_size, interval):
    """dense_patch_slices: window start tuples in monai's (row-major, first dimension slowest) order"""
    per_dim = []
    for img, roi, iv in zip(image_size, roi_size, interval):
        if iv == 0:
            num = 1
        else:
            cnt = int(math.ceil(float(img) / iv))
            first = next((d for d in range(cnt) if d * iv + roi >= img), None)
            num = first + 1 if first is not None else 1
        starts = []
        for idx in range(num):
            s = idx * iv
            s -= max(s + roi - img, 0)
            starts.append(s)
        per_dim.append(starts)
    out = [()]
    for starts in per_dim:
        out = [o + (s,) for o in out for s in starts]
    return out


def padding(image_size, roi_size):
    """symmetric zero padding of dimensions smaller than the window: (lo, hi) per dimension"""
    pads = []
    for img, roi in zip(image_size, roi_size):
        diff = max(roi - img, 0)
        pads.append((diff // 2, diff - diff // 2))
    return pads


def sliding_window_inference(inputs, roi_size, sw_batch_size, predictor, overlap=0.25):
    """inputs [B, Cin, H, W, D]; predictor maps a window batch to [n, Cout, h, w, d].  Constant blending."""
    B = inputs.shape[0]
    img0 = tuple(inputs.shape[2:])
    roi = tuple(int(r) if r and r > 0 else int(i) for r, i in zip(roi_size, img0))
    pads = padding(img0, roi)
    flat = []
    for lo, hi in reversed(pads):
        flat += [lo, hi]
    x = F.pad(inputs, flat, mode='constant', value=0.0)
    img = tuple(max(i, r) for i, r in zip(img0, roi))
    starts = patch_starts(img, roi, scan_interval(img, roi, overlap))
    nwin = len(starts)
    total = nwin * B
    out = cnt = None
    for g in range(0, total, sw_batch_size):
        idxs = list(range(g, min(g + sw_batch_size, total)))
        wins = []
        for idx in idxs:
            b, st = idx // nwin, starts[idx % nwin]
            wins.append(x[b:b + 1, :, st[0]:st[0] + roi[0], st[1]:st[1] + roi[1], st[2]:st[2] + roi[2]])
        seg = predictor(torch.cat(wins))
        if out is None:
            out = torch.zeros((B, seg.shape[1]) + img, dtype=torch.float32)
            cnt = torch.zeros((B, 1) + img, dtype=torch.float32)
        for k, idx in enumerate(idxs):
            b, st = idx // nwin, starts[idx % nwin]
            sl = (slice(b, b + 1), slice(None), slice(st[0], st[0] + roi[0]), slice(st[1], st[1] + roi[1]), slice(st[2], st[2] + roi[2]))
            out[sl] += seg[k:k + 1].float()
            cnt[sl] += 1.0
    out = out / cnt
    return out[:, :, pads[0][0]:pads[0][0] + img0[0], pads[1][0]:pads[1][0] + img0[1], pads[2][0]:pads[2][0] + img0[2]]


# ---------------------------------------------------------------------------------------------- metrics

def _class_rows(predict, target, class_index):
    p = predict.flatten(2).transpose(2, 1)[:, :, class_index]
    t = target.flatten(2).transpose(2, 1).squeeze(2)
    return p, t


def recall(predict, target, class_index=1, eps=1e-5):
    """criterions.py:280-311"""
    p, t = _class_rows(predict, target, class_index)
    return torch.mean((torch.sum(p * t, dim=-1) + eps) / (torch.sum(t, dim=-1) + eps))


def precision(predict, target, class_index=1, eps=1e-5):
    """criterions.py:348-379"""
    p, t = _class_rows(predict, target, class_index)
    return torch.mean((torch.sum(p * t, dim=-1) + eps) / (torch.sum(p, dim=-1) + eps))


def localization_loss(predict, target, class_index=1, eps=1e-6, mask_threshold=10):
    """criterions.py:179-241.  The loop over the three axes transposes only for i == 0 (a no-op, transpose(2, 2)) and flattens the
    un-transposed tensor otherwise, so every term is the profile along the first spatial axis; restated as written."""
    pr = predict[:, class_index].clone().unsqueeze(1)
    n_dim = pr.dim() - 2
    total = None
    for i in range(n_dim):
        if i != 0:
            dp, dl = pr.flatten(3), target.flatten(3)
        else:
            dp, dl = pr.transpose(2, i + 2).flatten(3), target.transpose(2, i + 2).flatten(3)
        dp = torch.sigmoid(torch.sum(dp, dim=-1) - mask_threshold)
        dl = torch.sigmoid(torch.sum(dl, dim=-1) - mask_threshold)
        cp = torch.cumsum(dp, dim=-1) / (torch.sum(dp, dim=-1, keepdim=True) + eps)
        ct = torch.cumsum(dl, dim=-1) / (torch.sum(dl, dim=-1, keepdim=True) + eps)
        term = 8 * torch.mean(torch.abs(cp - ct))
        total = term if total is None else total + term
    return total / n_dim


def keep_largest_component(predict):
    """inference_multi_classes.py:146-151: round, monai 0.7.0 KeepLargestConnectedComponent(applied_labels=[1, 2],
    independent=False, connectivity=3) (= skimage.measure.label(connectivity=3) on the union of the foreground channels, largest
    label by bincount/argmax, everything else cleared in those channels), then channel 0 = 1 - the rest.  scipy's labelling with a
    full 3x3x3 structure numbers components in the same raster order as skimage.  predict [B, C, H, W, D] torch tensor."""
    import numpy as np
    from scipy import ndimage
    out = torch.round(predict.float()).clone()
    for b in range(out.shape[0]):
        fg = (out[b, 1:].sum(0) > 0).numpy()
        lab, n = ndimage.label(fg, structure=np.ones((3, 3, 3), dtype=bool))
        if n > 0:
            keep = np.argmax(np.bincount(lab.ravel())[1:]) + 1
            drop = torch.from_numpy(fg & (lab != keep))
            for c in range(1, out.shape[1]):
                out[b, c][drop] = 0
        out[b, 0] = 1 - out[b, 1:].sum(0)
    return out
