"""Sliding-window whole-volume inference on the MI355X.

Mirrors the reference driver inference_embed_attn.py:92-185:
    predict = sliding_window_inference(images, (512, 512, depth), 4, model, overlap=0.6, sigma_scale=0)   # monai 0.7.0
    predict2 = (predict >= 0.5).float()
    DiceClassLoss / Recall / Precision / LocalizationLoss (predict2, masks)
with the same argument meaning.  Window scheduling is host integer logic; the windows, the votes of the model's eval
(one-hot arg-max) output and the metrics stay in HBM and go through the C-ABI (csrc/infer.hip).  No CPU fallback.
"""
import math

import torch

from . import _lib, ops
from .ops import _p, _s


def scan_interval(image_size, roi_size, overlap):
    """monai/inferers/utils.py::_get_scan_interval"""
    out = []
    for img, roi in zip(image_size, roi_size):
        if roi == img:
            out.append(int(roi))
        else:
            iv = int(roi * (1 - overlap))
            out.append(iv if iv > 0 else 1)
    return tuple(out)


def patch_starts(image_size, roi_size, interval):
    """monai/data/utils.py::dense_patch_slices: window starts, first dimension slowest"""
    per_dim = []
    for img, roi, iv in zip(image_size, roi_size, interval):
        if iv == 0:
            num = 1
        else:
            cnt = int(math.ceil(float(img) / iv))
            first = next((d for d in range(cnt) if d * iv + roi >= img), None)
            num = first + 1 if first is not None else 1
        per_dim.append([idx * iv - max(idx * iv + roi - img, 0) for idx in range(num)])
    out = [()]
    for starts in per_dim:
        out = [o + (s,) for o in out for s in starts]
    return out


def sliding_window_inference(inputs, roi_size, sw_batch_size, predictor, overlap=0.25):
    """inputs f32 [B, 1, H, W, D] on the GPU; predictor(windows [n, 1, h, w, d]) -> [n, C, h, w, d] (the eval-mode
    MaskTransUnet: channels-last one-hot exposed in the reference's shape).  Returns f32 [B, C, H, W, D], the per-voxel
    average of the window outputs (constant blending), exactly as monai's function does for mode="constant"."""
    if not inputs.is_cuda:
        raise _lib.LtuError('sliding_window_inference runs on the GPU only (no CPU fallback)')
    if inputs.dim() != 5 or inputs.shape[1] != 1:
        raise _lib.LtuError('inputs must be [B, 1, H, W, D]')
    B = inputs.shape[0]
    img0 = tuple(int(v) for v in inputs.shape[2:])
    roi = tuple(int(r) if r and r > 0 else i for r, i in zip(roi_size, img0))
    pad_lo = tuple(max(r - i, 0) // 2 for r, i in zip(roi, img0))
    img = tuple(max(i, r) for i, r in zip(img0, roi))
    starts = patch_starts(img, roi, scan_interval(img, roi, overlap))
    nwin, total = len(starts), len(starts) * B
    dev = inputs.device
    vol = inputs.to(torch.float32).contiguous()
    votes = count = None
    C = None
    for g in range(0, total, sw_batch_size):
        idxs = range(g, min(g + sw_batch_size, total))
        desc = torch.tensor([[idx // nwin, *starts[idx % nwin]] for idx in idxs], dtype=torch.int32).to(dev)
        n = len(idxs)
        win = torch.empty((n, 1) + roi, device=dev, dtype=torch.float32)
        _lib.call('ltu_window_gather', _p(vol), _p(win), _p(desc), n, img0[0], img0[1], img0[2], roi[0], roi[1], roi[2],
                  pad_lo[0], pad_lo[1], pad_lo[2], _s())
        seg = predictor(win)
        if seg.dim() != 5 or tuple(seg.shape[2:]) != roi or seg.shape[0] != n:
            raise _lib.LtuError(f'predictor returned {tuple(seg.shape)} for windows {(n, 1) + roi}')
        seg_cl = seg.permute(0, 2, 3, 4, 1)
        if seg_cl.dtype != torch.float32 or not seg_cl.is_contiguous():
            seg_cl = seg_cl.to(torch.float32).contiguous()
        if votes is None:
            C = seg.shape[1]
            votes = torch.zeros((B, C) + img, device=dev, dtype=torch.float32)
            count = torch.zeros((B,) + img, device=dev, dtype=torch.float32)
        _lib.call('ltu_vote_accumulate', _p(seg_cl), _p(votes), _p(count), _p(desc), n, img[0], img[1], img[2], roi[0], roi[1],
                  roi[2], C, _s())
    out = torch.empty((B, C) + img0, device=dev, dtype=torch.float32)
    _lib.call('ltu_vote_finalize', _p(votes), _p(count), _p(out), B, C, img0[0], img0[1], img0[2], img[0], img[1], img[2],
              pad_lo[0], pad_lo[1], pad_lo[2], _s())
    return out


METRIC_NAMES = ('DiceClassLoss', 'Recall', 'Precision', 'LocalizationLoss')


def evaluate(predict, masks, threshold=0.5, class_index=1):
    """The driver's metrics on predict2 = (predict >= threshold) against masks [B, 1, H, W, D] (0/1):
    returns {name: device scalar} for DiceClassLoss, Recall, Precision, LocalizationLoss (loss/criterions.py)."""
    if not predict.is_cuda:
        raise _lib.LtuError('evaluate runs on the GPU only (no CPU fallback)')
    B, C, H, W, D = predict.shape
    pred = predict.to(torch.float32).contiguous()
    tgt = masks.reshape(B, H, W, D).to(torch.uint8).contiguous()
    rows = torch.empty((B, 3, H), device=pred.device, dtype=torch.float32)
    values = torch.empty(4, device=pred.device, dtype=torch.float32)
    _lib.call('ltu_seg_metrics', _p(pred), _p(tgt), _p(rows), _p(values), B, C, class_index, H, W * D, float(threshold), _s())
    return {name: values[i] for i, name in enumerate(METRIC_NAMES)}


def keep_largest_component(predict, sweeps_per_check=8):
    """inference_multi_classes.py:146-151 on predict [B, C, H, W, D] (the blended votes): round, keep the largest 26-connected
    component of the foreground union (monai KeepLargestConnectedComponent(applied_labels=[1, 2], independent=False,
    connectivity=3)), channel 0 = 1 - the rest.  Returns a new tensor."""
    if not predict.is_cuda:
        raise _lib.LtuError('keep_largest_component runs on the GPU only (no CPU fallback)')
    B, C, H, W, D = predict.shape
    S = H * W * D
    out = torch.round(predict.to(torch.float32)).contiguous()
    dev = out.device
    labels = torch.empty(S, device=dev, dtype=torch.int32)
    for b in range(B):
        counts = torch.zeros(S + 1, device=dev, dtype=torch.int32)
        best = torch.zeros(1, device=dev, dtype=torch.int64)
        changed = torch.zeros(1, device=dev, dtype=torch.int32)
        args = (_p(out[b]), _p(labels), _p(counts), _p(best), _p(changed), C, H, W, D)
        _lib.call('ltu_keep_largest_component', *args, 0, _s())
        while True:
            changed.zero_()
            for _ in range(sweeps_per_check):
                _lib.call('ltu_keep_largest_component', *args, 1, _s())
            if changed.item() == 0:          # host check: an evaluation-time post-processing step, not on the training path
                break
        _lib.call('ltu_keep_largest_component', *args, 2, _s())
    return out


class GraphedPredictor:
    """The eval-mode forward for a fixed window batch captured once into a HIP graph and replayed per window batch: an eager
    forward is ~500 launches of ~35 us host time each, several times what the kernels need.  A short last batch is padded with
    copies of its first window (their outputs are dropped)."""

    def __init__(self, model, batch, roi, device):
        self.model, self.n = model, batch
        self.x = torch.zeros((batch, 1) + tuple(roi), device=device, dtype=torch.float32)
        self.ctx = ops.Context()          # own scratch arena: the graph bakes its addresses in, nobody else may move it
        side = torch.cuda.Stream(device=device)
        side.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(side), torch.no_grad(), ops.use(self.ctx):
            for _ in range(2):
                model(self.x)
        torch.cuda.current_stream(device).wait_stream(side)
        torch.cuda.synchronize(device)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, capture_error_mode='thread_local'), torch.no_grad(), ops.use(self.ctx):
            self.y = model(self.x)
        self.ctx.freeze()
        self.sig = tuple(p.data_ptr() for p in model.parameters())

    def __call__(self, win):
        if tuple(p.data_ptr() for p in self.model.parameters()) != self.sig:
            raise _lib.LtuError('GraphedPredictor: parameter storage moved since capture (build the predictor after .to() / '
                                'optimizer construction, or build a new one)')
        n = win.shape[0]
        self.x[:n].copy_(win)
        if n < self.n:
            self.x[n:].copy_(win[:1].expand(self.n - n, *win.shape[1:]))
        self.graph.replay()
        return self.y[:n]


def infer_volume(model, images, depth_size=32, roi_xy=512, sw_batch_size=4, overlap=0.6, graph=False):
    """one patient of inference_embed_attn.py:main: eval-mode model, (roi_xy, roi_xy, depth_size) windows, overlap 0.6.
    graph=True (or a GraphedPredictor built earlier) replays the forward from a captured HIP graph."""
    was_training = model.training
    model.eval()
    try:
        with torch.no_grad():
            predictor = model
            if isinstance(graph, GraphedPredictor):
                predictor = graph
            elif graph:
                roi = tuple(r if r and r > 0 else i for r, i in zip((roi_xy, roi_xy, depth_size), images.shape[2:]))
                predictor = GraphedPredictor(model, sw_batch_size, roi, images.device)
            return sliding_window_inference(images, (roi_xy, roi_xy, depth_size), sw_batch_size, predictor, overlap=overlap)
    finally:
        model.train(was_training)
