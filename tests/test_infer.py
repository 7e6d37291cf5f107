"""Sliding-window inference driver: oracle pinned by golden metrics; HIP path vs oracle."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import infer as O  # noqa: E402
from oracle import losses as OL  # noqa: E402


# ---------------------------------------------------------------------------------------------- CPU: oracle
def test_oracle_metrics_golden(golden_dir):
    G = np.load(os.path.join(golden_dir, 'metrics.npz'))
    for tag in ('a', 'b'):
        p, t = torch.from_numpy(G[f'{tag}_predict']), torch.from_numpy(G[f'{tag}_target']).long()
        got = [OL.dice_class(p, t), O.recall(p, t), O.precision(p, t), O.localization_loss(p, t.float())]
        for v, ref in zip(got, G[f'{tag}_values']):
            assert abs(v.item() - ref) <= 1e-6 * max(1.0, abs(ref))


@pytest.mark.parametrize('img,roi,overlap', [((512, 512, 40), (512, 512, 32), 0.6), ((70, 33, 16), (32, 32, 16), 0.6),
                                              ((20, 20, 8), (32, 16, 8), 0.25), ((9, 9, 9), (4, 4, 4), 0.0)])
def test_oracle_window_schedule(img, roi, overlap):
    """every voxel of the (padded) image is covered, windows stay inside, the last window touches the end"""
    pimg = tuple(max(i, r) for i, r in zip(img, roi))
    starts = O.patch_starts(pimg, roi, O.scan_interval(pimg, roi, overlap))
    cover = np.zeros(pimg, dtype=np.int32)
    for s in starts:
        assert all(0 <= a and a + r <= i for a, r, i in zip(s, roi, pimg))
        cover[s[0]:s[0] + roi[0], s[1]:s[1] + roi[1], s[2]:s[2] + roi[2]] += 1
    assert cover.min() >= 1
    assert len(set(starts)) == len(starts)
    if img == (512, 512, 40):          # the reference configuration: 1 x 1 x 2 windows of depth 32 over a 40-slice scan
        assert starts == [(0, 0, 0), (0, 0, 8)]


def test_oracle_sliding_window_identity():
    """with an identity-like predictor the blended output equals the input (constant weights average equal values)"""
    x = torch.randn(2, 1, 19, 11, 7)
    out = O.sliding_window_inference(x, (8, 8, 4), 3, lambda w: torch.cat((w, -w), 1), overlap=0.5)
    assert torch.allclose(out[:, :1], x, atol=1e-6) and torch.allclose(out[:, 1:], -x, atol=1e-6)
    small = torch.randn(1, 1, 5, 6, 3)      # smaller than the window: symmetric zero padding, cropped again
    out = O.sliding_window_inference(small, (8, 8, 4), 2, lambda w: torch.cat((w, 2 * w), 1), overlap=0.6)
    assert out.shape == (1, 2, 5, 6, 3) and torch.allclose(out[:, :1], small, atol=1e-6)


def test_host_schedule_matches_oracle():
    from lintransunet_amd import infer as P
    for img, roi, ov in [((512, 512, 40), (512, 512, 32), 0.6), ((70, 33, 16), (32, 32, 16), 0.6), ((9, 9, 9), (4, 4, 4), 0.1)]:
        assert P.scan_interval(img, roi, ov) == O.scan_interval(img, roi, ov)
        assert P.patch_starts(img, roi, P.scan_interval(img, roi, ov)) == O.patch_starts(img, roi, O.scan_interval(img, roi, ov))


# ---------------------------------------------------------------------------------------------- GPU
DEV = 'cuda'


def _onehot_predictor(w):
    """a deterministic stand-in model: 3 classes from thresholds of the window intensities, one-hot, channels-last memory"""
    cls = (w[:, 0] > 0.3).long() + (w[:, 0] > 1.0).long()
    oh = torch.nn.functional.one_hot(cls, 3).to(torch.float32)       # [n, h, w, d, C]
    return oh.permute(0, 4, 1, 2, 3)


@pytest.mark.gpu
@pytest.mark.parametrize('shape,roi,sw,overlap', [((2, 1, 37, 21, 12), (16, 16, 8), 4, 0.6), ((1, 1, 10, 40, 6), (16, 16, 8), 3, 0.6),
                                                   ((1, 1, 33, 33, 9), (32, 32, 8), 2, 0.25)])
def test_sliding_window_matches_oracle(shape, roi, sw, overlap):
    from lintransunet_amd import infer as P
    g = torch.Generator().manual_seed(5)
    x = torch.randn(shape, generator=g)
    ref = O.sliding_window_inference(x, roi, sw, _onehot_predictor, overlap=overlap)
    got = P.sliding_window_inference(x.to(DEV), roi, sw, _onehot_predictor, overlap=overlap)
    assert got.shape == ref.shape
    assert torch.equal(got.cpu(), ref)          # votes and counts are small integers: exact


@pytest.mark.gpu
def test_metrics_golden(golden_dir):
    from lintransunet_amd import infer as P
    G = np.load(os.path.join(golden_dir, 'metrics.npz'))
    for tag in ('a', 'b'):
        p, t = torch.from_numpy(G[f'{tag}_predict']).to(DEV), torch.from_numpy(G[f'{tag}_target']).to(DEV)
        vals = P.evaluate(p, t, threshold=0.5)
        for name, ref in zip(P.METRIC_NAMES, G[f'{tag}_values']):
            assert abs(vals[name].item() - ref) <= 1e-5 * max(1.0, abs(ref)), name


@pytest.mark.gpu
def test_infer_volume_with_model():
    """the real eval-mode network as the predictor: output is a per-voxel average of one-hot votes"""
    from lintransunet_amd import infer as P
    from lintransunet_amd.model import get_model_dict
    torch.manual_seed(3)
    model = get_model_dict('MaskTransUnet')([8, 8, 8, 16, 32], [20, 12, 9, 10, 6], [False, True, True, True, True], 1, 2).to(DEV)
    x = torch.randn(1, 1, 48, 32, 40, device=DEV)
    out = P.infer_volume(model, x, depth_size=32, roi_xy=32, sw_batch_size=2, overlap=0.6)
    assert out.shape == (1, 2, 48, 32, 40)
    assert torch.allclose(out.sum(1), torch.ones_like(out[:, 0]), atol=1e-6)      # votes of a one-hot predictor sum to 1
    vals = P.evaluate(out, (torch.rand(1, 1, 48, 32, 40, device=DEV) > 0.5))
    assert all(torch.isfinite(v) for v in vals.values())
    assert model.training                                                         # mode restored
    out_g = P.infer_volume(model, x, depth_size=32, roi_xy=32, sw_batch_size=2, overlap=0.6, graph=True)
    assert torch.equal(out_g, out)                                                # graph replay of the same kernels


@pytest.mark.gpu
def test_config5_window_batch_vs_oracle():
    """BASELINE config 5 at its real window size: a 512x512x40 scan = two 512x512x32 windows (overlap 0.6, sw batch 4) through
    the window gather / vote accumulate / finalize kernels, against oracle/infer.py with the same stand-in predictor: exact"""
    from lintransunet_amd import infer as P
    g = torch.Generator().manual_seed(6)
    x = torch.randn((1, 1, 512, 512, 40), generator=g)
    ref = O.sliding_window_inference(x, (512, 512, 32), 4, _onehot_predictor, overlap=0.6)
    got = P.sliding_window_inference(x.to(DEV), (512, 512, 32), 4, _onehot_predictor, overlap=0.6)
    assert torch.equal(got.cpu(), ref)


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.float32])
def test_config5_model_golden(golden_dir, dtype):
    """the real eval-mode network (reference channel / ROI configuration) over the same 512x512x40 scan: class-1 votes against the
    vectors produced by the REFERENCE model under the oracle's window driver (tests/golden/make_golden.py infer512).  The network
    has random weights, so 48 % of the voxels are foreground and many sit near the arg-max tie: fp32 flips a few (8e-6 observed).
    bf16 storage is not compared voxel-wise here: with random weights a bf16-sized perturbation moves an ROI box edge by one cell
    (see tests/test_gpu_model.py) and a quarter of the near-tie voxels flip; bf16 inference is gated on a trained model by
    tests/test_heldout.py."""
    from lintransunet_amd import infer as P
    from lintransunet_amd.model import get_model_dict
    from oracle import net as O_net, seedgen
    G = np.load(os.path.join(golden_dir, 'infer512.npz'))
    cfg = O_net.NetConfig()
    model = get_model_dict('MaskTransUnet')(cfg.num_layers, cfg.roi_size_list, cfg.is_roi_list, 1, 2, act_dtype=dtype)
    model.load_state_dict(seedgen.seeded_params(O_net.param_shapes(cfg), 700), strict=True)
    model = model.to(DEV)
    x = seedgen.seeded_volume((1, 1, 512, 512, 40), 701).to(DEV)
    out = P.infer_volume(model, x, depth_size=32, roi_xy=512, sw_batch_size=4, overlap=0.6)
    assert out.shape == (1, 2, 512, 512, 40)
    votes = (out[0, 1] * 2).round().to(torch.uint8).cpu()
    ref = torch.from_numpy(G['votes2'])
    mism = (votes != ref).float().mean().item()
    print(f'[config5 {dtype}] voxels whose vote differs from the reference: {mism:.3e}')
    assert mism <= (2e-4 if dtype == torch.float32 else 3e-2)
    # Dice between the thresholded volumes (what inference_embed_attn.py:146-150 scores)
    a, b = (votes >= 1).double(), (ref >= 1).double()
    dice = (2 * (a * b).sum() / (a.sum() + b.sum())).item()
    assert dice >= (1 - 2e-4 if dtype == torch.float32 else 0.97)


def _blobs(seed, shape=(2, 3, 20, 18, 14)):
    g = torch.Generator().manual_seed(seed)
    B, C, H, W, D = shape
    p = torch.zeros(shape)
    for b in range(B):
        for k in range(5):                     # a few boxes of either foreground class, some touching only diagonally
            h0, w0, d0 = [int(torch.randint(0, n - 5, (1,), generator=g)) for n in (H, W, D)]
            c = 1 + k % 2
            p[b, c, h0:h0 + 2 + k, w0:w0 + 3, d0:d0 + 2 + k % 3] = 1.0
        p[b, 2] *= (1 - p[b, 1])
    p[:, 0] = 1 - p[:, 1] - p[:, 2]
    return p * 0.9 + 0.04                      # blended votes: rounding recovers the one-hot


def test_oracle_keep_largest_component():
    p = _blobs(1)
    out = O.keep_largest_component(p)
    assert torch.all(out.sum(1) == 1)
    fg_in, fg_out = torch.round(p)[:, 1:].sum(1) > 0, out[:, 1:].sum(1) > 0
    assert torch.all(fg_in | ~fg_out) and 0 < fg_out.sum() < fg_in.sum()


@pytest.mark.gpu
def test_keep_largest_component_matches_oracle():
    from lintransunet_amd import infer as P
    for seed in (1, 2, 3):
        p = _blobs(seed)
        assert torch.equal(P.keep_largest_component(p.to(DEV)).cpu(), O.keep_largest_component(p))
    empty = torch.zeros(1, 3, 6, 5, 4); empty[:, 0] = 1
    assert torch.equal(P.keep_largest_component(empty.to(DEV)).cpu(), empty)
