"""GPU parity of the whole MaskTransUnet step against the vectors produced by the reference itself
(tests/golden/make_golden.py).  Gates: fp32 storage: max|diff|/max|ref| <= 1e-3 on the output
and every mask, |dDice| <= 1e-4, boxes bit-exact, gradient norms within 1e-2 relative (fp32 atomics / summation order);
bf16 storage: Dice gate only."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import net as O_net          # noqa: E402
from oracle import seedgen               # noqa: E402
from oracle import step as O_step        # noqa: E402

DEV = 'cuda'
SMALL = dict(num_layers=[8, 8, 8, 16, 32], roi_size_list=[20, 12, 9, 10, 6])


def exact_zero_grad(key):
    """Parameters whose gradient is mathematically zero: the bias of a conv feeding InstanceNorm (the norm removes
    any per-channel constant) and the key-projection bias (softmax over tokens is shift invariant).  Reference and HIP
    path both produce rounding noise there (1e-3 .. 1e-8), so a relative comparison is meaningless."""
    if not key.endswith('.bias'):
        return False
    return any(t in key for t in ('.conv1.', '.conv2.', 'input_block', 'down_embed', 'up_embed', 'W_x.0', 'W_g.0',
                                  'self_attn.linears.1.'))


def grads_agree(a, b, key):
    """Two runs of the same step.  fp32 atomics make the InstanceNorm sums differ in the last bits between runs; an
    activation that sits within 1e-7 of zero can then flip its LeakyReLU/ReLU branch, which changes ONE term of a weight
    gradient (visible on layers with few rows, e.g. the 896-voxel token embedding).  Hence a relative L2 criterion."""
    if exact_zero_grad(key):
        return (a - b).abs().max().item() <= 1e-5
    return (a - b).double().norm().item() <= 5e-3 * max(b.double().norm().item(), 1e-6)


def rel_err(a, b):
    a, b = a.detach().double().cpu(), torch.as_tensor(np.asarray(b)).double()
    return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-30)


def build(cfg, wseed, dtype=torch.float32, dropout=0.0):
    from lintransunet_amd.model import get_model_dict
    model = get_model_dict('MaskTransUnet')(cfg.num_layers, cfg.roi_size_list, cfg.is_roi_list, cfg.dim_input, cfg.dim_output,
                                            dropout=dropout, act_dtype=dtype)
    model.load_state_dict(seedgen.seeded_params(O_net.param_shapes(cfg), wseed), strict=True)
    return model.to(DEV).train()


def run(cfg, size, batch, wseed, dtype=torch.float32):
    from lintransunet_amd import train
    model = build(cfg, wseed, dtype)
    x = seedgen.seeded_volume((batch, 1) + size, wseed + 1).to(DEV)
    label = seedgen.seeded_label((batch, 1) + size, wseed + 2).to(DEV)
    predict, masks = model(x)
    totals, named = train.deep_supervision_loss(predict, masks, label, O_step.dynamic_weights(0))
    torch.autograd.backward(totals, [torch.ones_like(t) for t in totals])
    torch.cuda.synchronize()
    return model, x, label, predict, masks, totals, named


def test_model_small_fp32(golden_dir):
    G = np.load(os.path.join(golden_dir, 'model_small.npz'))
    cfg = O_net.NetConfig(**SMALL)
    model, x, label, predict, masks, totals, named = run(cfg, (32, 32, 32), 2, 100)
    assert predict.shape == (2, 2, 32, 32, 32)
    for i, b in enumerate(model.last_boxes):
        assert torch.equal(b.cpu(), torch.from_numpy(G[f'box{i}'])), f'box{i}'
    assert rel_err(predict, G['out']) <= 1e-3
    for i, m in enumerate(masks):
        assert rel_err(m, G[f'mask{i}']) <= 1e-3, f'mask{i}'
    total = sum(t.item() for t in totals)
    assert abs(total - float(G['total'])) <= 1e-4 * max(1.0, abs(float(G['total'])))
    lv = G['level_losses']
    for lvl, vals in enumerate(named):
        got = [v.item() for v in vals.values()]
        assert np.allclose(got, lv[lvl], rtol=1e-4, atol=1e-5), (lvl, got, lv[lvl])
    from lintransunet_amd import losses as L
    dice = L.DiceClassLoss()(predict.detach(), label).item()
    assert abs(dice - float(G['dice'])) <= 1e-4
    norms = dict(zip(G['grad_keys'], G['grad_norms']))
    sd = dict(model.named_parameters())
    worst = 0.0
    for k, n in norms.items():
        got = sd[k].grad.double().norm().item()
        if exact_zero_grad(k):
            assert got <= 1e-2, k
            continue
        worst = max(worst, abs(got - n) / max(n, 1e-3))
    assert worst <= 1e-2, worst
    for k in G['nograd_keys']:
        assert sd[str(k)].grad is None
    for k in G.files:
        if k.startswith('grad::'):
            assert rel_err(sd[k[6:]].grad, G[k]) <= 5e-3, k
    model.eval()
    with torch.no_grad():
        onehot = model(x)
    assert onehot.shape == predict.shape
    assert abs(int(onehot[:, 1].sum().item()) - int(G['onehot_fg_count'])) <= 2


# full128 / full96: the BASELINE patch sizes with the reference's channel / ROI configuration (configs 3 and 2), vectors from the
# reference itself (tests/golden/make_golden.py fullsize)
SAMPLED = [('small_wide', SMALL, (64, 96, 16), 1, 200), ('full32', {}, (32, 32, 32), 1, 300),
           ('full128', {}, (128, 128, 128), 1, 500), ('full96', {}, (96, 96, 96), 2, 600),
           ('win512', {}, (512, 512, 32), 1, 800)]       # the reference's own training crop / inference window


@pytest.mark.parametrize('tag,cfgkw,size,batch,wseed', SAMPLED)
def test_model_sampled_fp32(golden_dir, tag, cfgkw, size, batch, wseed):
    G = np.load(os.path.join(golden_dir, f'model_{tag}.npz'))
    cfg = O_net.NetConfig(**cfgkw)
    model, x, label, predict, masks, totals, named = run(cfg, size, batch, wseed)
    for i, b in enumerate(model.last_boxes):
        assert torch.equal(b.cpu(), torch.from_numpy(G[f'box{i}'])), f'box{i}'
    flat = predict.detach().cpu().flatten()
    assert rel_err(flat[torch.from_numpy(G['out_idx'])], G['out_sample']) <= 1e-3
    total = sum(t.item() for t in totals)
    assert abs(total - float(G['total'])) <= 1e-4 * max(1.0, abs(float(G['total'])))
    from lintransunet_amd import losses as L
    assert abs(L.DiceClassLoss()(predict.detach(), label).item() - float(G['dice'])) <= 1e-4
    norms = dict(zip(G['grad_keys'], G['grad_norms']))
    sd = dict(model.named_parameters())
    worst = max(abs(sd[k].grad.double().norm().item() - n) / max(n, 1e-3) for k, n in norms.items()
                if not exact_zero_grad(k))
    assert worst <= 1e-2, worst
    lv = G['level_losses']
    for lvl, vals in enumerate(named):
        got = [v.item() for v in vals.values()]
        assert np.allclose(got, lv[lvl], rtol=1e-4, atol=1e-5), (lvl, got, lv[lvl])
    for i, m in enumerate(masks):
        ref = G[f'mask{i}']
        mm = m.detach().cpu()
        got = mm.numpy() if mm.numel() <= 70000 else mm.flatten()[:: max(1, mm.numel() // 4096)].numpy()
        assert rel_err(torch.from_numpy(np.ascontiguousarray(got)), ref) <= 1e-3, f'mask{i}'


# bf16 storage at the BASELINE sizes.  Gate: Dice ("bf16: Dice gate only").  The reference's own bf16-autocast vs fp32
# pair differs by 8e-5 in Dice at 32^3; BF16_DICE_TOL is what this path is held to against the fp32 reference
# (observed 7e-5 .. 1.8e-4 on these random-weight cases, where the Dice loss is 0.84 .. 0.92, i.e. nearly degenerate; the gate on
# a trained model and a held-out volume is tests/test_heldout.py).
# The ROI boxes are step functions of the coarser mask (get_mask_boundary2 thresholds it at 0.5, Unet_3Dblock.py:738-739).  With
# random weights the masks hover around 0.5, so a bf16-sized perturbation can move a box edge by one cell; every activation behind
# that ROI bridge then differs by O(1) although each op is exact to bf16 (a per-op trace (round 2) shows the op-by-op picture: 2e-2
# drift up to the first moved box, 3e-1 right after it).  The element-wise gate therefore applies when the boxes agree.
BF16_DICE_TOL = 3e-4
BF16_DICE_TOL_MOVED_BOX = 1e-3
# Gradient norms of the bf16 step against the reference's 600 fp32 norms (same fixture the fp32 path is held to at 1e-2), when the
# boxes agree.  This random-weight network amplifies bf16 rounding of the activations (a whole-gradient rel-L2 of ~0.15 between
# bf16 and fp32 storage, test_layer_tail_matches_op_by_op), so single tensors move by a few per cent; gates are the measured
# distribution with margin: median, 95th percentile and maximum of |norm - ref| / max(ref, 1e-3) over the ~590 tensors.
BF16_GRADNORM_TOL = dict(median=2e-2, p95=0.12, max=0.3)      # observed: median <= 9.3e-3, p95 <= 7.7e-2, max <= 0.13


def bf16_gradnorm_check(model, G, tag):
    norms = dict(zip(G['grad_keys'], G['grad_norms']))
    sd = dict(model.named_parameters())
    rel = []
    for k, n in norms.items():
        if exact_zero_grad(k):
            continue
        got = sd[k].grad.double().norm().item()
        assert np.isfinite(got), k
        rel.append((abs(got - n) / max(n, 1e-3), k))
    rel.sort()
    med, p95, mx = rel[len(rel) // 2][0], rel[int(len(rel) * 0.95)][0], rel[-1][0]
    print(f'[bf16 {tag}] gradient norms vs reference over {len(rel)} tensors: median {med:.2e}, p95 {p95:.2e}, max {mx:.2e} ({rel[-1][1]})')
    assert med <= BF16_GRADNORM_TOL['median'] and p95 <= BF16_GRADNORM_TOL['p95'] and mx <= BF16_GRADNORM_TOL['max'], (med, p95, mx, rel[-3:])


@pytest.mark.parametrize('tag,cfgkw,size,batch,wseed', [c for c in SAMPLED if c[0] in ('full128', 'full96', 'full32', 'win512')])
def test_model_sampled_bf16(golden_dir, tag, cfgkw, size, batch, wseed):
    G = np.load(os.path.join(golden_dir, f'model_{tag}.npz'))
    cfg = O_net.NetConfig(**cfgkw)
    model, x, label, predict, masks, totals, named = run(cfg, size, batch, wseed, torch.bfloat16)
    from lintransunet_amd import losses as L
    dice = L.DiceClassLoss()(predict.detach(), label).item()
    total = sum(t.item() for t in totals)
    flat = predict.detach().cpu().flatten()
    ref = torch.from_numpy(G['out_sample']).double()
    rel_l2 = ((flat[torch.from_numpy(G['out_idx'])].double() - ref).norm() / ref.norm()).item()
    print(f'[bf16 {tag}] dice {dice:.6f} vs {float(G["dice"]):.6f} (d {dice - float(G["dice"]):+.2e}), total {total:.6f} vs '
          f'{float(G["total"]):.6f}, sampled rel-L2 {rel_l2:.2e}')
    moved = 0.0
    for i, b in enumerate(model.last_boxes):
        moved = max(moved, (b.cpu() - torch.from_numpy(G[f'box{i}'])).abs().max().item())
    assert moved <= 1.0
    assert abs(dice - float(G['dice'])) <= (BF16_DICE_TOL if moved == 0 else BF16_DICE_TOL_MOVED_BOX)
    assert abs(total - float(G['total'])) <= 1e-2 * abs(float(G['total']))
    if moved == 0:
        assert rel_l2 <= 3e-2
        bf16_gradnorm_check(model, G, tag)
    assert all(torch.isfinite(p.grad).all() for p in model.parameters() if p.grad is not None)


def test_model_small_bf16_dice(golden_dir):
    """bf16 activation storage: Dice gate only (the reference's own bf16-vs-fp32 forward differs by 1.4e-2 rel-L2)"""
    G = np.load(os.path.join(golden_dir, 'model_small.npz'))
    cfg = O_net.NetConfig(**SMALL)
    model, x, label, predict, masks, totals, named = run(cfg, (32, 32, 32), 2, 100, torch.bfloat16)
    from lintransunet_amd import losses as L
    dice = L.DiceClassLoss()(predict.detach(), label).item()
    print(f'[bf16 small] dice {dice:.6f} vs {float(G["dice"]):.6f}')
    assert abs(dice - float(G['dice'])) <= BF16_DICE_TOL
    assert all(torch.isfinite(p.grad).all() for p in model.parameters() if p.grad is not None)


def test_dropout_training_runs():
    """p = 0.3 (the reference default): finite loss/grads, different masks on consecutive steps"""
    from lintransunet_amd import train
    cfg = O_net.NetConfig(**SMALL)
    model = build(cfg, 100, dropout=0.3)
    x = seedgen.seeded_volume((1, 1, 32, 32, 32), 1).to(DEV)
    label = seedgen.seeded_label((1, 1, 32, 32, 32), 2).to(DEV)
    t1, _ = train.train_step(model, x, label, O_step.dynamic_weights(0))
    t2, _ = train.train_step(model, x, label, O_step.dynamic_weights(0))
    a, b = sum(t.item() for t in t1), sum(t.item() for t in t2)
    assert np.isfinite(a) and np.isfinite(b) and a != b
    assert all(torch.isfinite(p.grad).all() for p in model.parameters() if p.grad is not None)


def test_fused_gradient_path_matches_autograd_path():
    """GradReducer installs `_ltu_grad` targets: the wgrad kernels then accumulate straight into the flat gradient
    buffer.  Same step, same weights: both routes must give the same gradients (and the arena must not corrupt them)."""
    from lintransunet_amd import train
    cfg = O_net.NetConfig(**SMALL)
    x = seedgen.seeded_volume((2, 1, 32, 32, 32), 11).to(DEV)
    label = seedgen.seeded_label((2, 1, 32, 32, 32), 12).to(DEV)
    w = O_step.dynamic_weights(0)
    ref = build(cfg, 100)
    train.train_step(ref, x, label, w)
    fused = build(cfg, 100)
    red = train.GradReducer(fused, bucket_mb=0.5, unused=train.UNUSED_PARAMETERS)
    for _ in range(2):                       # second step: buffers and arena are recycled
        red.zero_grad()
        train.train_step(fused, x, label, w, reducer=red)
    torch.cuda.synchronize()
    pr, pf = dict(ref.named_parameters()), dict(fused.named_parameters())
    for k, p in pr.items():
        if p.grad is None:
            continue
        assert grads_agree(pf[k].grad, p.grad, k), k


def test_graphed_step_matches_eager():
    """HIP-graph replay of the captured step: same gradients as the eager step; the device-resident step counter keeps
    dropout masks fresh across replays"""
    from lintransunet_amd import train
    cfg = O_net.NetConfig(**SMALL)
    x = seedgen.seeded_volume((2, 1, 32, 32, 32), 21).to(DEV)
    label = seedgen.seeded_label((2, 1, 32, 32, 32), 22).to(DEV)
    w = O_step.dynamic_weights(0)
    ref = build(cfg, 100)
    train.train_step(ref, x, label, w)
    m = build(cfg, 100)
    red = train.GradReducer(m, bucket_mb=0.5, unused=train.UNUSED_PARAMETERS)
    g = train.GraphedStep(m, x, label, w, red)
    for _ in range(2):
        totals, _ = g(x, label)
    torch.cuda.synchronize()
    pr, pm = dict(ref.named_parameters()), dict(m.named_parameters())
    for k, p in pr.items():
        if p.grad is None:
            continue
        assert grads_agree(pm[k].grad, p.grad, k), k
    # with dropout the replays must differ (fresh masks) although the captured seeds are frozen
    md = build(cfg, 100, dropout=0.3)
    redd = train.GradReducer(md, bucket_mb=0.5, unused=train.UNUSED_PARAMETERS)
    gd = train.GraphedStep(md, x, label, w, redd)
    a = sum(t.item() for t in gd(x, label)[0])
    b = sum(t.item() for t in gd(x, label)[0])
    assert np.isfinite(a) and np.isfinite(b) and a != b


def _flat_grads(model):
    return {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}


def test_graphed_step_accumulation_and_weights():
    """utils/utils_3D_embed_full.py:85-91: `step_times` micro-steps accumulate (each loss divided by step_times), the buckets are
    zeroed on the first and reduced on the last; the per-epoch level weights (train3D.py:122-137) are a device tensor, so a
    captured graph follows `set_weights` without re-capture"""
    from lintransunet_amd import train
    cfg = O_net.NetConfig(**SMALL)
    xs = [seedgen.seeded_volume((1, 1, 32, 32, 32), 41 + i).to(DEV) for i in range(2)]
    ls = [seedgen.seeded_label((1, 1, 32, 32, 32), 51 + i).to(DEV) for i in range(2)]
    w0, w1 = O_step.dynamic_weights(0), O_step.dynamic_weights(40)
    assert w0 != w1

    def eager(w):
        m = build(cfg, 100)
        red = train.GradReducer(m, bucket_mb=0.5, unused=train.UNUSED_PARAMETERS)
        red.zero_grad()
        tot = 0.0
        for j in range(2):
            t, _ = train.train_step(m, xs[j], ls[j], w, step_times=2, reducer=red, reduce=(j == 1))
            tot += sum(v.item() for v in t)
        torch.cuda.synchronize()
        return _flat_grads(m), tot

    m = build(cfg, 100)
    red = train.GradReducer(m, bucket_mb=0.5, unused=train.UNUSED_PARAMETERS)
    g = train.GraphedStep(m, xs[0], ls[0], w0, red, step_times=2)
    for w in (w0, w1, w0):                     # weights change between optimizer steps; graphs are reused
        g.set_weights(w)
        tot = 0.0
        for j in range(2):
            t, _ = g(xs[j], ls[j], micro=j)
            tot += sum(v.item() for v in t)
        torch.cuda.synchronize()
        ref, ref_tot = eager(w)
        got = _flat_grads(m)
        assert abs(tot - ref_tot) <= 1e-5 * abs(ref_tot)
        for k, r in ref.items():
            assert grads_agree(got[k], r, k), k
    assert set(g.graphs) == {(True, False), (False, True), (True, True)}


def test_graphed_step_recaptures_when_storage_moves():
    """an optimizer built after the capture re-homes every parameter into its flat buffers (optim.FusedAdamW): the next replay must
    notice (data_ptr check) and capture again instead of reading the old, never-updated weights"""
    from lintransunet_amd import train, optim
    cfg = O_net.NetConfig(**SMALL)
    x = seedgen.seeded_volume((1, 1, 32, 32, 32), 61).to(DEV)
    lab = seedgen.seeded_label((1, 1, 32, 32, 32), 62).to(DEV)
    w = O_step.dynamic_weights(0)
    m = build(cfg, 100)
    red = train.GradReducer(m, bucket_mb=0.5, unused=train.UNUSED_PARAMETERS)
    g = train.GraphedStep(m, x, lab, w, red)
    l0 = sum(t.item() for t in g(x, lab)[0])
    first = g.graphs[(True, True)][0]
    opt = optim.FusedAdamW(red, lr=1e-2)            # moves p.data
    opt.step()
    l1 = sum(t.item() for t in g(x, lab)[0])
    assert g.graphs[(True, True)][0] is not first       # re-captured
    ref = build(cfg, 100)
    ref.load_state_dict(m.state_dict())
    lr = sum(t.item() for t in train.train_step(ref, x, lab, w)[0])
    assert abs(l1 - lr) <= 1e-4 * abs(lr) and abs(l1 - l0) > 1e-6      # the replay saw the updated weights


def test_rebucket_in_gradient_ready_order():
    """buckets re-assigned in the order the gradients become ready (train.GradReducer.rebucket): the decoder tail comes first, the
    encoder last (registration order puts the bottleneck transformer and the coarse decoder stages early although they finish
    late); gradients are unchanged and a captured step notices the moved storage"""
    from lintransunet_amd import train
    cfg = O_net.NetConfig(**SMALL)
    x = seedgen.seeded_volume((1, 1, 32, 32, 32), 91).to(DEV)
    lab = seedgen.seeded_label((1, 1, 32, 32, 32), 92).to(DEV)
    w = O_step.dynamic_weights(0)
    m = build(cfg, 100)
    red = train.GradReducer(m, bucket_mb=0.25, unused=train.UNUSED_PARAMETERS)
    g = train.GraphedStep(m, x, lab, w, red)
    g(x, lab)
    torch.cuda.synchronize()
    before = _flat_grads(m)
    names = {id(p): n for n, p in m.named_parameters()}
    red.zero_grad()
    train.train_step(m, x, lab, w, reducer=red)          # an eager step records the ready order
    order = [names[id(p)] for p in red.ready_order]
    assert order[0].startswith('decode.final_block') and order[-1].startswith('encode.')
    first_enc = min(i for i, n in enumerate(order) if n.startswith('encode.'))
    assert all(n.startswith('encode.') for n in order[first_enc:])          # the encoder finishes last, as one block
    assert order.index('decode.bridge_list.4.transformer.layers.0.linear1.weight') > order.index('decode.bridge_list.1.transformer.layers.0.linear1.weight')
    red.rebucket()
    assert [names[id(p)] for b in red.buckets for p in b][:len(order)] == order
    g(x, lab)                                                 # gradient storage moved: captured again
    torch.cuda.synchronize()
    after = _flat_grads(m)
    for k, v in before.items():
        assert grads_agree(after[k], v, k), k


def test_contexts_isolate_captured_arenas():
    """a GraphedPredictor captured while the scratch arena was small keeps working after a larger GraphedStep has been built and
    run, and an evaluation between a training forward and its backward does not disturb that backward"""
    from lintransunet_amd import train, infer as P
    cfg = O_net.NetConfig(**SMALL)
    m = build(cfg, 100)
    win = seedgen.seeded_volume((1, 1, 32, 32, 32), 71).to(DEV)
    m.eval()
    pred = P.GraphedPredictor(m, 1, (32, 32, 32), win.device)
    with torch.no_grad():
        want = m(win).clone()
    m.train()
    assert torch.equal(pred(win), want)
    x = seedgen.seeded_volume((2, 1, 64, 64, 32), 72).to(DEV)          # a larger training step: more scratch
    lab = seedgen.seeded_label((2, 1, 64, 64, 32), 73).to(DEV)
    w = O_step.dynamic_weights(0)
    red = train.GradReducer(m, bucket_mb=0.5, unused=train.UNUSED_PARAMETERS)
    g = train.GraphedStep(m, x, lab, w, red)
    g(x, lab)
    torch.cuda.synchronize()
    assert torch.equal(pred(win), want)                                 # the predictor's arena was not moved or overwritten
    # eval between forward and backward of an eager step
    ref = build(cfg, 100)
    train.train_step(ref, x, lab, w)
    m2 = build(cfg, 100)
    with ops_use_default():
        predict, masks = m2(x)
        totals, _ = train.deep_supervision_loss(predict, masks, lab, w)
        m2.eval()
        with torch.no_grad():
            m2(win)
        m2.train()
        torch.autograd.backward(totals, [torch.ones_like(t) for t in totals])
    torch.cuda.synchronize()
    pr, pm = dict(ref.named_parameters()), dict(m2.named_parameters())
    for k, p in pr.items():
        if p.grad is not None:
            assert grads_agree(pm[k].grad, p.grad, k), k


def ops_use_default():
    from lintransunet_amd import ops
    ctx = ops.Context()
    ctx.begin_step(torch.device(DEV))
    return ops.use(ctx)


@pytest.mark.parametrize('dropout', [0.0, 0.3])
def test_layer_tail_matches_op_by_op(dropout):
    """the row-block chain kernel (csrc/tlayer.hip: out projection, LayerNorm, FFN, LayerNorm of a transformer layer in one launch,
    used on the small token levels) against the op-by-op path on the same weights, inputs and dropout masks: the full channel
    configuration at 32^3 x 2 runs it at d = 128 (28 704 tokens) and d = 256 (5 376 / 2 160 tokens).
    Yardstick: this random-weight network amplifies bf16 rounding noise, so both bf16 paths sit ~0.15 (relative L2 of the whole
    gradient) away from the fp32-storage run with the same masks; the two paths must agree far better than that."""
    from lintransunet_amd import train, ops
    cfg = O_net.NetConfig()
    x = seedgen.seeded_volume((2, 1, 32, 32, 32), 81).to(DEV)
    label = seedgen.seeded_label((2, 1, 32, 32, 32), 82).to(DEV)
    w = O_step.dynamic_weights(0)
    calls = {'n': 0}
    orig = ops._LayerTail.forward

    def counting(*a, **k):
        calls['n'] += 1
        return orig(*a, **k)

    def run(dtype, tail):
        ops.USE_LAYER_TAIL = tail
        torch.manual_seed(99)
        m = build(cfg, 300, dtype, dropout=dropout)
        t, _ = train.train_step(m, x, label, w)
        torch.cuda.synchronize()
        return (sum(v.item() for v in t), {k: p.grad.double() for k, p in m.named_parameters() if p.grad is not None and not exact_zero_grad(k)},
                [b.clone() for b in m.last_boxes])

    def dist(a, b):
        num = sum(((a[k] - b[k]) ** 2).sum().item() for k in b) ** 0.5
        return num / sum((b[k] ** 2).sum().item() for k in b) ** 0.5

    ops._LayerTail.forward = staticmethod(counting)
    try:
        la, ga, ba = run(torch.bfloat16, True)
        assert calls['n'] == 24                 # 3 ROI transformers x 8 layers (the 16-token bottleneck keeps the op-by-op path)
        lb, gb, bb = run(torch.bfloat16, False)
        assert calls['n'] == 24
        lf, gf, bf_ = run(torch.float32, False)
        ops.FUSE_ATTN_APPLY = False             # chain kernels behind the stand-alone attention phase B (same arithmetic)
        lc_, gc, bc = run(torch.bfloat16, True)
    finally:
        ops.USE_LAYER_TAIL = True
        ops.FUSE_ATTN_APPLY = True
        ops._LayerTail.forward = staticmethod(orig)
    assert all(torch.equal(p, q) for p, q in zip(ba, bc))
    assert abs(la - lc_) <= 1e-6 * abs(lc_), (la, lc_)
    d_fuse = dist(ga, gc)
    print(f'[layer tail] attention phase B inside the chain kernel vs stand-alone: gradient rel-L2 {d_fuse:.2e}')
    assert d_fuse <= 1e-3
    assert all(torch.equal(p, q) for p, q in zip(ba, bb))
    assert abs(la - lb) <= 2e-4 * abs(lb), (la, lb)
    d_paths, d_ref = dist(ga, gb), dist(gb, gf)
    print(f'[layer tail, dropout {dropout}] loss {la:.6f} vs {lb:.6f} (fp32 {lf:.6f}); gradient rel-L2 chain vs op-by-op {d_paths:.2e}, '
          f'op-by-op vs fp32 storage {d_ref:.2e}, chain vs fp32 storage {dist(ga, gf):.2e}')
    assert d_paths <= 0.3 * d_ref and abs(dist(ga, gf) - d_ref) <= 0.1 * d_ref


def test_smoke_entry():
    import __graft_entry__
    __graft_entry__.smoke()


def test_model_multiclass_fp32(golden_dir):
    """dim_output = 3, the multi-class step of utils/utils_3D_multi_class.py:68-102 (one-hot targets from the
    max-pooled integer labels, criterion weights 10 / 1 / 2) against vectors from the reference's own model + multi_criterions."""
    from lintransunet_amd import train
    from lintransunet_amd import losses as L
    G = np.load(os.path.join(golden_dir, 'model_multi_small.npz'))
    cfg = O_net.NetConfig(dim_output=3, **SMALL)
    model = build(cfg, 400)
    x = seedgen.seeded_volume((2, 1, 32, 32, 32), 401).to(DEV)
    label = seedgen.seeded_label((2, 1, 32, 32, 32), 402, n_classes=3).to(DEV)
    predict, masks = model(x)
    assert predict.shape == (2, 3, 32, 32, 32)
    specs = train.level_specs(5, ('CrossEntroLoss', 'DiceClassLoss', 'DiceClassLoss2'), criterion_weight=[10, 1, 2])
    totals, named = train.deep_supervision_loss(predict, masks, label, O_step.dynamic_weights(0), specs=specs)
    torch.autograd.backward(totals, [torch.ones_like(t) for t in totals])
    torch.cuda.synchronize()
    assert rel_err(predict, G['out']) <= 1e-3
    for i, m in enumerate(masks):
        assert rel_err(m, G[f'mask{i}']) <= 1e-3, f'mask{i}'
    total = sum(t.item() for t in totals)
    assert abs(total - float(G['total'])) <= 1e-4 * max(1.0, abs(float(G['total'])))
    lv = G['level_losses']
    for lvl, vals in enumerate(named):
        got = [vals[n].item() for n in ('CrossEntroLoss', 'DiceClassLoss', 'DiceClassLoss2')]
        assert np.allclose(got, lv[lvl], rtol=1e-4, atol=1e-5), (lvl, got, lv[lvl])
    assert abs(L.DiceClassLoss()(predict.detach(), label).item() - float(G['dice1'])) <= 1e-4
    assert abs(L.DiceClassLoss2()(predict.detach(), label).item() - float(G['dice2'])) <= 1e-4
    norms = dict(zip(G['grad_keys'], G['grad_norms']))
    sd = dict(model.named_parameters())
    worst = 0.0
    for k, n in norms.items():
        got = sd[k].grad.double().norm().item()
        if exact_zero_grad(k):
            assert got <= 1e-2, k
            continue
        worst = max(worst, abs(got - n) / max(n, 1e-3))
    assert worst <= 1e-2, worst


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_model_multiclass_128(golden_dir, dtype):
    """BASELINE config 4 at its stated size: 128^3, 3 labels, the reference channel / ROI configuration, the multi-class step of
    utils/utils_3D_multi_class.py:68-102 (CE + Dice_1 + Dice_2 weighted 10 / 1 / 2) against vectors from the reference's own model +
    multi_criterions (tests/golden/make_golden.py multi128)"""
    from lintransunet_amd import train
    from lintransunet_amd import losses as L
    G = np.load(os.path.join(golden_dir, 'model_multi128.npz'))
    cfg = O_net.NetConfig(dim_output=3)
    model = build(cfg, 900, dtype)
    x = seedgen.seeded_volume((1, 1, 128, 128, 128), 901).to(DEV)
    label = seedgen.seeded_label((1, 1, 128, 128, 128), 902, n_classes=3).to(DEV)
    predict, masks = model(x)
    specs = train.level_specs(5, ('CrossEntroLoss', 'DiceClassLoss', 'DiceClassLoss2'), criterion_weight=[10, 1, 2])
    totals, named = train.deep_supervision_loss(predict, masks, label, O_step.dynamic_weights(0), specs=specs)
    torch.autograd.backward(totals, [torch.ones_like(t) for t in totals])
    torch.cuda.synchronize()
    total = sum(t.item() for t in totals)
    d1 = L.DiceClassLoss()(predict.detach(), label).item()
    d2 = L.DiceClassLoss2()(predict.detach(), label).item()
    flat = predict.detach().cpu().flatten()[torch.from_numpy(G['out_idx'])]
    if dtype == torch.float32:
        assert rel_err(flat, G['out_sample']) <= 1e-3
        assert abs(total - float(G['total'])) <= 1e-4 * abs(float(G['total']))
        assert abs(d1 - float(G['dice1'])) <= 1e-4 and abs(d2 - float(G['dice2'])) <= 1e-4
        lv = G['level_losses']
        for lvl, vals in enumerate(named):
            got = [vals[n].item() for n in ('CrossEntroLoss', 'DiceClassLoss', 'DiceClassLoss2')]
            assert np.allclose(got, lv[lvl], rtol=1e-4, atol=1e-5), (lvl, got, lv[lvl])
        norms = dict(zip(G['grad_keys'], G['grad_norms']))
        sd = dict(model.named_parameters())
        worst = max(abs(sd[k].grad.double().norm().item() - n) / max(n, 1e-3) for k, n in norms.items() if not exact_zero_grad(k))
        assert worst <= 1e-2, worst
    else:
        print(f'[bf16 multi128] dice1 {d1:.6f} vs {float(G["dice1"]):.6f}, dice2 {d2:.6f} vs {float(G["dice2"]):.6f}, total {total:.5f} vs {float(G["total"]):.5f}')
        assert abs(d1 - float(G['dice1'])) <= BF16_DICE_TOL_MOVED_BOX and abs(d2 - float(G['dice2'])) <= BF16_DICE_TOL_MOVED_BOX
        assert abs(total - float(G['total'])) <= 1e-2 * abs(float(G['total']))
        assert all(torch.isfinite(p.grad).all() for p in model.parameters() if p.grad is not None)
        ref = torch.from_numpy(G['out_sample']).double()
        rel_l2 = ((flat.double() - ref).norm() / ref.norm()).item()
        print(f'[bf16 multi128] sampled rel-L2 {rel_l2:.2e}')
        if rel_l2 <= 3e-2:          # the fixture holds no boxes: bf16-level agreement of the output means none has moved
            bf16_gradnorm_check(model, G, 'multi128')


def test_model_multiclass_bf16(golden_dir):
    """the 3-label configuration in bf16 storage (BASELINE config 4 runs it): 4C = 12 final-conv channels are padded to 16 for the
    bf16 GEMMs; outputs, losses and Dice stay within bf16 distance of the fp32 vectors generated by the reference"""
    from lintransunet_amd import train
    from lintransunet_amd import losses as L
    G = np.load(os.path.join(golden_dir, 'model_multi_small.npz'))
    cfg = O_net.NetConfig(dim_output=3, **SMALL)
    model = build(cfg, 400, dtype=torch.bfloat16)
    x = seedgen.seeded_volume((2, 1, 32, 32, 32), 401).to(DEV)
    label = seedgen.seeded_label((2, 1, 32, 32, 32), 402, n_classes=3).to(DEV)
    predict, masks = model(x)
    assert predict.shape == (2, 3, 32, 32, 32) and (predict.sum(1) - 1).abs().max().item() <= 1e-5
    specs = train.level_specs(5, ('CrossEntroLoss', 'DiceClassLoss', 'DiceClassLoss2'), criterion_weight=[10, 1, 2])
    totals, named = train.deep_supervision_loss(predict, masks, label, O_step.dynamic_weights(0), specs=specs)
    torch.autograd.backward(totals, [torch.ones_like(t) for t in totals])
    torch.cuda.synchronize()
    ref = torch.from_numpy(G['out']).double()
    assert ((predict.detach().double().cpu() - ref).norm() / ref.norm()).item() <= 3e-2        # rel-L2 (single voxels move by up to 0.1)
    total = sum(t.item() for t in totals)
    assert abs(total - float(G['total'])) <= 2e-2 * abs(float(G['total']))
    assert abs(L.DiceClassLoss()(predict.detach(), label).item() - float(G['dice1'])) <= 1e-2
    assert abs(L.DiceClassLoss2()(predict.detach(), label).item() - float(G['dice2'])) <= 1e-2
    for k, p in model.named_parameters():
        if p.grad is not None:
            assert torch.isfinite(p.grad).all(), k


def test_full_size_properties():
    """BASELINE size (128^3, 2 patches per GPU, the reference's channel / ROI configuration, bf16 storage): besides the golden
    comparisons above (test_model_sampled_fp32 / _bf16 [full128]), size-independent properties: class probabilities sum to 1
    everywhere, every op is per-sample (swapping the two patches swaps outputs, masks and ROI boxes), and the gradient is linear
    in the loss scale."""
    from lintransunet_amd import train
    from lintransunet_amd.model import get_model_dict
    torch.manual_seed(7)
    model = get_model_dict('MaskTransUnet')([16, 32, 64, 128, 256], [100, 65, 40, 25, 10], [False, True, True, True, True], 1, 2,
                                            dropout=0.0, act_dtype=torch.bfloat16).to(DEV).train()
    x = seedgen.seeded_volume((2, 1, 128, 128, 128), 31).to(DEV)
    label = seedgen.seeded_label((2, 1, 128, 128, 128), 32).to(DEV)
    weights = O_step.dynamic_weights(0)

    def step(xx, ll, scale):
        for p in model.parameters():
            p.grad = None
        predict, masks = model(xx)
        boxes = [b.clone() for b in model.last_boxes]
        totals, _ = train.deep_supervision_loss(predict, masks, ll, weights, scale=scale)
        torch.autograd.backward(totals, [torch.ones_like(t) for t in totals])
        return predict.detach(), [m.detach() for m in masks], boxes, sum(t.item() for t in totals)

    p0, m0, b0, l0 = step(x, label, 1.0)
    g0 = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
    assert p0.shape == (2, 2, 128, 128, 128) and torch.isfinite(p0).all()
    assert (p0.sum(1) - 1).abs().max().item() <= 1e-5
    for m in m0:
        assert (m.sum(1) - 1).abs().max().item() <= 1e-5
    # per-sample structure
    p1, m1, b1, l1 = step(x.flip(0), label.flip(0), 1.0)
    assert (p1.flip(0) - p0).abs().max().item() <= 2e-3
    for a, b in zip(m1, m0):
        assert (a.flip(0) - b).abs().max().item() <= 2e-3
    for a, b in zip(b1, b0):
        assert torch.equal(a.flip(0), b)
    assert abs(l1 - l0) <= 1e-3 * abs(l0)
    # linearity in the loss scale
    _, _, _, l2 = step(x, label, 0.5)
    assert abs(l2 - 0.5 * l0) <= 1e-3 * abs(l0)
    worst, worst_key = 0.0, None
    for k, p in model.named_parameters():
        if p.grad is None or exact_zero_grad(k):
            continue
        n0 = g0[k].double().norm().item()
        dev = (p.grad.double() - 0.5 * g0[k].double()).norm().item() / max(n0, 1e-9)
        if dev > worst:
            worst, worst_key = dev, (k, n0)
    assert worst <= 2e-2, (worst, worst_key)


def test_step_is_reproducible():
    """two identical training steps (bf16 storage, reference channel configuration, 32^3 x 2) give bit-identical gradients: the
    reductions of the step are two-stage with a fixed order - the level losses included, whose fp32 atomics used to move the last
    bit of every loss coefficient from run to run, which bf16 rounding downstream turned into 1e-2 of some gradients - and so is
    the weight / bias gradient of the positional depthwise conv (the last atomics of the step)."""
    from lintransunet_amd import train
    cfg = O_net.NetConfig()
    x = seedgen.seeded_volume((2, 1, 32, 32, 32), 81).to(DEV)
    label = seedgen.seeded_label((2, 1, 32, 32, 32), 82).to(DEV)
    w = O_step.dynamic_weights(0)

    def run():
        torch.manual_seed(99)
        m = build(cfg, 300, torch.bfloat16, dropout=0.0)
        t, _ = train.train_step(m, x, label, w)
        torch.cuda.synchronize()
        return [v.item() for v in t], {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}

    l0, g0 = run()
    for _ in range(3):
        l, g = run()
        assert l == l0
        for k in g0:
            assert torch.equal(g[k], g0[k]), k
