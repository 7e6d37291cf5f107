"""Optimizer / schedule / checkpoint shell.  The oracle is what the reference itself uses: torch.optim.AdamW and
torch.optim.lr_scheduler.ReduceLROnPlateau (CPU)."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def test_plateau_schedule_matches_torch():
    from lintransunet_amd.optim import ReduceLROnPlateau

    class Dummy:
        def __init__(self):
            self.param_groups = [{'lr': 1e-4}]

    g = torch.Generator().manual_seed(0)
    losses = (1.0 + 0.05 * torch.randn(120, generator=g)).cumsum(0).div(torch.arange(1, 121)).tolist()
    losses = [v if i % 17 else v * 0.8 for i, v in enumerate(losses)]
    p = torch.nn.Parameter(torch.zeros(1))
    topt = torch.optim.AdamW([p], lr=1e-4)
    tsch = torch.optim.lr_scheduler.ReduceLROnPlateau(topt, mode='min', factor=0.8, patience=5, threshold=1e-2, cooldown=1, min_lr=1e-7)
    mine = Dummy()
    msch = ReduceLROnPlateau(mine, mode='min', factor=0.8, patience=5, threshold=1e-2, cooldown=1, min_lr=1e-7)
    for v in losses + [5.0] * 80:
        tsch.step(v)
        msch.step(v)
        assert abs(mine.param_groups[0]['lr'] - topt.param_groups[0]['lr']) <= 1e-18
    assert mine.param_groups[0]['lr'] < 1e-4


@pytest.mark.gpu
def test_fused_adamw_matches_torch(tmp_path):
    from lintransunet_amd import train, optim
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(37, 19), torch.nn.Linear(19, 5)).cuda()
    ref = torch.nn.Sequential(torch.nn.Linear(37, 19), torch.nn.Linear(19, 5))
    ref.load_state_dict({k: v.cpu() for k, v in model.state_dict().items()})
    reducer = train.GradReducer(model, bucket_mb=0.001)          # several buckets, odd sizes (tail path of the kernel)
    opt = optim.FusedAdamW(reducer, lr=1e-2)
    topt = torch.optim.AdamW(ref.parameters(), lr=1e-2)
    for it in range(7):
        x = torch.randn(11, 37)
        opt.zero_grad(); topt.zero_grad()
        model(x.cuda()).square().mean().backward()
        ref(x).square().mean().backward()
        opt.step(); topt.step()
        for (k, a), (_, b) in zip(model.state_dict().items(), ref.state_dict().items()):
            assert torch.allclose(a.cpu(), b, rtol=2e-5, atol=2e-7), (it, k)
    ck = optim.BestCheckpoint(str(tmp_path))
    assert ck.update(model, 0.5, 0.7, opt) and not ck.update(model, 0.6, 0.1, opt)
    sd = torch.load(os.path.join(str(tmp_path), 'temp_model.pt'))
    ref.load_state_dict(sd, strict=True)                        # a plain state_dict, loadable without this package
    extra = torch.load(os.path.join(str(tmp_path), 'temp_model.extra.pt'), weights_only=False)
    assert extra['optimizer']['step'] == 7


@pytest.mark.gpu
def test_training_loop_with_graph_and_fused_adamw():
    """forward + loss + backward replayed from the captured HIP graph, AdamW on the flat buckets in between: the graph keeps
    reading the updated master weights (same storage) and the loss goes down on a fixed batch"""
    from lintransunet_amd import train, optim
    from lintransunet_amd.model import get_model_dict
    from oracle import seedgen, step as O_step
    torch.manual_seed(1)
    model = get_model_dict('MaskTransUnet')([8, 8, 8, 16, 32], [20, 12, 9, 10, 6], [False, True, True, True, True], 1, 2,
                                            dropout=0.0).cuda().train()
    reducer = train.GradReducer(model, unused=train.UNUSED_PARAMETERS)
    opt = optim.FusedAdamW(reducer, lr=2e-3, weight_decay=0.0)
    x = seedgen.seeded_volume((2, 1, 32, 32, 32), 11).cuda()
    lab = seedgen.seeded_label((2, 1, 32, 32, 32), 12).cuda()
    step = train.GraphedStep(model, x, lab, O_step.dynamic_weights(0), reducer)
    w0 = model.decode.final_block.weight.detach().clone()
    hist = []
    for _ in range(8):
        totals, _ = step(x, lab)
        hist.append(sum(t.item() for t in totals))
        opt.step()
    assert hist[-1] < hist[0] * 0.98, hist
    assert (model.decode.final_block.weight.detach() - w0).abs().max().item() > 0
