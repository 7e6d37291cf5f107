"""Per-call timing of every C-ABI entry point during one training step (GPU box).

Wraps lintransunet_amd._lib.call with HIP events (synchronising after each call, so numbers are
serialised kernel times) and prints the calls aggregated by (entry point, integer arguments).

    python tools/profile_ops.py [--dtype bf16] [--size 128] [--batch 2] [--top 60]
"""
import argparse
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from lintransunet_amd import _lib, train            # noqa: E402
from lintransunet_amd.model import get_model_dict   # noqa: E402
import bench                                         # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--dtype', default='bf16')
    ap.add_argument('--size', type=int, default=128)
    ap.add_argument('--batch', type=int, default=2)
    ap.add_argument('--top', type=int, default=60)
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    act = torch.bfloat16 if args.dtype == 'bf16' else torch.float32
    torch.manual_seed(0)
    model = get_model_dict('MaskTransUnet')([16, 32, 64, 128, 256], [100, 65, 40, 25, 10], [False, True, True, True, True], 1, 2,
                                            dropout=0.3, act_dtype=act).to(dev).train()
    x, lab = bench.synthetic_batch(args.batch, (args.size,) * 3, 5, dev)
    weights = train.get_dynamic_weight(1)[0]
    reducer = train.GradReducer(model, unused=train.UNUSED_PARAMETERS)      # fused gradient buffers: the grouped weight-gradient path
    train.train_step(model, x, lab, weights, reducer=reducer)          # warm-up
    torch.cuda.synchronize()

    stats = collections.defaultdict(lambda: [0, 0.0])
    raw = _lib.call

    def timed(name, *a):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        raw(name, *a)
        e1.record()
        e1.synchronize()
        key = (name, tuple(v for v in a if isinstance(v, int) and not isinstance(v, bool) and abs(v) < (1 << 31))[:14])
        s = stats[key]
        s[0] += 1
        s[1] += e0.elapsed_time(e1)

    _lib.call = timed
    import lintransunet_amd.ops as ops
    ops._lib.call = timed
    reducer.zero_grad()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    train.train_step(model, x, lab, weights, reducer=reducer)
    t1.record(); t1.synchronize()
    total = sum(v[1] for v in stats.values())
    print(f'step wall (serialised) {t0.elapsed_time(t1):.1f} ms; sum of timed C calls {total:.1f} ms; {sum(v[0] for v in stats.values())} calls')
    by_name = collections.defaultdict(float)
    for (name, _), v in stats.items():
        by_name[name] += v[1]
    for name, ms in sorted(by_name.items(), key=lambda kv: -kv[1]):
        print(f'  {ms:8.2f} ms  {name}')
    # the four token transformers: calls whose integer arguments contain the level's token count (B * N) or per-sample count N
    print('--- transformer levels (event-timed eager calls: each includes ~5 us of launch overhead)')
    for M, N, d in ((114816, 57408, 128), (21504, 10752, 256), (8640, 4320, 256), (1024, 512, 256)):
        ms = sum(v[1] for (name, ints), v in stats.items() if (M in ints or N in ints) and name.startswith(('ltu_linear', 'ltu_layer_tail', 'ltu_linattn', 'ltu_layernorm', 'ltu_gelu')))
        n = sum(v[0] for (name, ints), v in stats.items() if (M in ints or N in ints) and name.startswith(('ltu_linear', 'ltu_layer_tail', 'ltu_linattn', 'ltu_layernorm', 'ltu_gelu')))
        print(f'  {M:7d} tokens x {d}: {ms:6.2f} ms in {n} calls')
    print('--- by call signature')
    for (name, ints), (n, ms) in sorted(stats.items(), key=lambda kv: -kv[1][1])[:args.top]:
        print(f'{ms:8.3f} ms  x{n:<3d} {ms / n * 1e3:8.1f} us  {name} {ints}')


if __name__ == '__main__':
    main()
