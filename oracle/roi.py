"""Oracle: dynamic-ROI box finder and the separable warp index maps (test infrastructure).

Restates model/Unet_3Dblock.py:
  * cumulative-ratio quantiles of a 1-D occupancy histogram   (37-49)
  * per-sample box with the pad / shrink rules                  (821-873)
  * forward piecewise-linear index map (image -> ROI grid)      (51-64)
  * inverse piecewise-linear index map (ROI grid -> image)      (66-82)
  * crop-and-warp / warp-back through 2-D bilinear sampling     (985-1039, 1080-1117)
Layout here is the reference's: [B, C, H, W, D] with D contiguous.
"""
import torch
import torch.nn.functional as F


def roi_geometry(roi_size: int):
    """Fixed ROI grid sizes derived from one `roi_size_list` entry (Unet_3Dblock.py:695-715)."""
    h_roi = roi_size
    w_roi = int(roi_size * 0.6)
    eval_h = int(1.2 * roi_size)
    eval_w = int(eval_h * 0.6)
    return dict(h_roi=h_roi, w_roi=w_roi, eval_h=eval_h, eval_w=eval_w,
                min_h=eval_h // 2, min_w=eval_w // 2)


def occupancy_quantiles(hist: torch.Tensor, thr: float = 0.001):
    """(lo, hi, median) positions of a 1-D histogram (Unet_3Dblock.py:37-49).

    Empty histogram -> (mid-1, mid+1, mid) with mid = len/2 as floats.
    Otherwise searchsorted on cumsum/sum at thr (left), 1-thr (right), 0.5 (right).
    """
    total = hist.sum()
    if total == 0:
        mid = torch.tensor(hist.shape[0] / 2)
        return mid - 1, mid + 1, mid
    ratio = torch.cumsum(hist, 0) / total
    lo = torch.searchsorted(ratio, thr, right=False)
    hi = torch.searchsorted(ratio, 1 - thr, right=True)
    med = torch.searchsorted(ratio, 0.5, right=True)
    return lo, hi, med


def _fit_extent(lo, hi, center, full: int, min_len: int):
    """Pad a too-small / shrink a too-large extent around `center` (Unet_3Dblock.py:847-871).

    Both rules test the ORIGINAL size; when both fire the second one wins.
    """
    size = hi - lo
    zero = torch.tensor(0)
    top = torch.tensor(full)
    out_lo, out_hi = lo, hi
    if size < min_len:
        out_lo = torch.maximum(center - min_len / 2, zero)
        out_hi = torch.minimum(center + min_len / 2, top)
    if size > (full - min_len):
        out_lo = torch.maximum(center - (full - min_len) / 2, zero)
        out_hi = torch.minimum(center + (full - min_len) / 2, top)
    return out_lo, out_hi


def find_boxes(fg_bool: torch.Tensor, min_h: int, min_w: int) -> torch.Tensor:
    """Boolean foreground [B,1,H,W,D] -> float boxes [B,6] = (x0,y0,0,x1,y1,D-1) (Unet_3Dblock.py:821-873)."""
    B = fg_bool.shape[0]
    H, W, D = fg_bool.shape[-3:]
    rows = fg_bool.sum(dim=(3, 4)).reshape(B, H)   # int64 counts per h
    cols = fg_bool.sum(dim=(2, 4)).reshape(B, W)   # int64 counts per w
    box = torch.zeros(B, 6, dtype=torch.float32)
    for b in range(B):
        x_lo, x_hi, x_c = occupancy_quantiles(rows[b])
        y_lo, y_hi, y_c = occupancy_quantiles(cols[b])
        # the reference stores lo/hi into a float32 row first and derives the sizes from it
        x_lo_f = torch.as_tensor(x_lo).to(torch.float32)
        x_hi_f = torch.as_tensor(x_hi).to(torch.float32)
        y_lo_f = torch.as_tensor(y_lo).to(torch.float32)
        y_hi_f = torch.as_tensor(y_hi).to(torch.float32)
        x0, x1 = _fit_extent(x_lo_f, x_hi_f, x_c, H, min_h)
        y0, y1 = _fit_extent(y_lo_f, y_hi_f, y_c, W, min_w)
        box[b, 0], box[b, 3] = x0, x1
        box[b, 1], box[b, 4] = y0, y1
        box[b, 2], box[b, 5] = 0, D - 1
    return box


def index_map_fwd(x0, x1, span: int, roi: int, eval_roi: int) -> torch.Tensor:
    """Normalised source coordinate for each of `eval_roi` ROI samples (Unet_3Dblock.py:51-64).

    x0,x1: [B,1] box edges; span = (image length - 1).  Slope k2 inside the box, k1 outside.
    """
    idx = torch.arange(0, eval_roi, dtype=torch.float32)
    k2 = (x1 - x0) / (roi - 1)
    k1 = (span - x1 + x0) / (eval_roi - roi)
    pos = idx * k2 + x0 * (1 - k2 / k1)
    below = pos <= x0
    alt = pos * (k1 / k2) + x0 * (1 - k1 / k2)
    pos[below] = alt[below]
    above = pos >= x1
    alt = pos * (k1 / k2) + x1 * (1 - k1 / k2)
    pos[above] = alt[above]
    return pos * 2. / span - 1


def index_map_back(x0, x1, span: int, roi: int, eval_roi: int) -> torch.Tensor:
    """Normalised ROI coordinate for each of span+1 image samples (Unet_3Dblock.py:66-82)."""
    idx = torch.arange(0, span + 1, dtype=torch.float32)
    k2 = roi / (x1 - x0)
    k1 = (eval_roi - roi) / (span - x1 + x0)
    p0 = x0 * k1
    p1 = eval_roi - (span - x1) * k1
    pos = idx * k2 + p0 * (1 - k2 / k1)
    below = pos <= p0
    alt = pos * (k1 / k2) + p0 * (1 - k1 / k2)
    pos[below] = alt[below]
    above = pos >= p1
    alt = pos * (k1 / k2) + p1 * (1 - k1 / k2)
    pos[above] = alt[above]
    return pos * 2 / eval_roi - 1


def _sample_planes(x: torch.Tensor, gx: torch.Tensor, gy: torch.Tensor) -> torch.Tensor:
    """Separable 2-D bilinear sampling of every depth slice (Unet_3Dblock.py:1010-1039 / 1101-1117).

    x [B,C,H,W,D]; gx [B,Eh], gy [B,Ew] normalised coordinates -> [B,C,Eh,Ew,D].
    """
    B, C, H, W, D = x.shape
    Eh, Ew = gx.shape[1], gy.shape[1]
    gx4 = gx[:, None, :, None].expand(B, D, Eh, Ew).flatten(0, 1)
    gy4 = gy[:, None, None, :].expand(B, D, Eh, Ew).flatten(0, 1)
    grid = torch.stack([gy4, gx4], dim=-1)
    planes = x.permute(0, 4, 1, 2, 3).flatten(0, 1)
    out = F.grid_sample(planes, grid.to(planes.dtype), align_corners=True)
    return out.reshape(B, D, C, Eh, Ew).permute(0, 2, 3, 4, 1)


def warp_to_roi(x: torch.Tensor, box: torch.Tensor, geo: dict) -> torch.Tensor:
    """Crop-and-warp the feature map into the fixed ROI grid (Unet_3Dblock.py:985-1039)."""
    H, W = x.shape[2], x.shape[3]
    x0, y0, _, x1, y1, _ = torch.split(box, 1, dim=1)
    gx = index_map_fwd(x0, x1, H - 1, geo['h_roi'], geo['eval_h'])
    gy = index_map_fwd(y0, y1, W - 1, geo['w_roi'], geo['eval_w'])
    return _sample_planes(x, gx, gy)


def warp_from_roi(like: torch.Tensor, roi: torch.Tensor, box: torch.Tensor, geo: dict) -> torch.Tensor:
    """Warp the processed ROI grid back onto the image lattice (Unet_3Dblock.py:1080-1117)."""
    H, W = like.shape[2], like.shape[3]
    x0, y0, _, x1, y1, _ = torch.split(box, 1, dim=1)
    gx = index_map_back(x0, x1, H - 1, geo['h_roi'], geo['eval_h'])
    gy = index_map_back(y0, y1, W - 1, geo['w_roi'], geo['eval_w'])
    return _sample_planes(roi, gx, gy)
