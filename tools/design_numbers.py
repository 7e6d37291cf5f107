"""Regenerates the measured blocks of DESIGN.md (between the AUTO markers) from profiles/r05_bench.json."""
import json, os, re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
p = os.path.join(ROOT, 'DESIGN.md')
s = open(p).read()
d = json.loads([l for l in open(os.path.join(ROOT, 'profiles', 'r05_bench.json')) if l.startswith('{')][0])
fam = {f['family']: f for f in d['roofline']['families']}
lk = d['roofline']['largest_kernel']
head = f"""`bench.py`: 128³ single-channel patches, 2 per GPU (BASELINE config 3 per-GPU share), bf16 storage, dropout 0.3, fwd + loss +
bwd, no optimizer step, inputs resident in HBM, step replayed from the captured graphs (linear segments on the compute stream +
weight-gradient batches on a side stream, §4a); `value` = patches/s of the whole job.
**{d['value']:.1f} patches/s on one MI355X in the committed profile (`profiles/r05_bench.json`: {d['ms_per_step']:.2f} ms per step, 20 timed
replays after 5; the boxes of this round gave 12.35-12.75 ms for the same code, so every change was judged by interleaved runs on ONE
box, `tools/ab_env.sh` / `tools/ms.sh`); round 4: 160.5, round 3: 137.2, round 2: 124.3, round 1: 96.3**; whole-step mixed
roofline (SURVEY §8d: 2.39 ms per patch at 100 %) ⇒ `step_frac` = {d['roofline']['step_frac']:.3f}.
`roofline` (chain kernels): {d['roofline']['avg_launch_ms'] * 1e3:.1f} µs per launch for {d['roofline']['algorithmic_bytes_per_launch'] / 1e6:.1f} MB algorithmic ⇒ {d['roofline']['achieved'] / 1e3:.2f} TB/s =
**{d['roofline']['frac']:.2f} of the HBM peak**; PMC traffic {(d['roofline']['traffic'] or 0) / 1e6:.1f} MB per launch.
`roofline.largest_kernel` (grouped projection weight gradients, {lk['launches']} launches, {lk['algorithmic_bytes_per_step'] / 1e9:.2f} GB of operands per step): {lk['ms_alone']:.2f} ms alone
at the side-stream width ⇒ {lk['achieved'] / 1e3:.2f} TB/s = {lk['frac']:.2f}; {lk['ms_in_step'] or 0:.2f} ms inside the step ⇒ {lk['frac_in_step'] or 0:.2f}; PMC traffic {(lk['traffic'] or 0) / 1e9:.2f} GB per step
= {(lk['traffic'] or 0) / lk['algorithmic_bytes_per_step']:.2f}× algorithmic (the three q|k|v column tiles and the row tiles of linear2 re-read their operands; the fat-tile kernel that reads
them once is finding 46).
`cpu_baseline`: the oracle, fp32, dropout on, 128³ B = 1, {d['cpu_baseline']['cores']} host threads, 1 warm-up + 2 timed steps: {d['cpu_baseline']['value']:.3f} patches/s ({d['cpu_baseline']['sample'].split('(')[-1].rstrip(')')})."""
r3 = {'transformer': 8.32, 'conv3': 3.86, 'instnorm': 1.19, 'resample': 0.49, 'other': 0.39, 'dwconv': 0.32}       # round 4, alone
names = {'transformer': 'transformer layers (projections, LayerNorm, GELU, linear attention, attention gates)',
         'conv3': '3×3×3 convolutions (level convs, strided / sub-pixel embedding convs; forward + both gradients)',
         'instnorm': 'InstanceNorm + LeakyReLU + residual + dropout', 'resample': 'ROI warp / un-warp, trilinear upsampling, adjoints',
         'other': 'window embedding, heads, losses, label pyramid, weight preparation, fills', 'dwconv': 'positional depthwise conv'}
rows = "| family | ms alone | ms in the step | C-ABI calls | bound | algorithmic | achieved (alone) | fraction of peak | round 4, alone |\n|---|---|---|---|---|---|---|---|---|\n"
ins = lambda f: f"{f['ms_in_step']:.2f}" if f.get('ms_in_step') is not None else '-'
for k in ('transformer', 'conv3', 'instnorm', 'resample', 'other', 'dwconv'):
    f = fam[k]
    if 'bound' in f:
        alg = f"{f['algorithmic_gb']:.1f} GB" if f['bound'] == 'hbm' else f"{f['algorithmic_gflop']:.0f} GFLOP (nominal)"
        ach = f"{f['achieved'] / 1e3:.2f} TB/s" if f['bound'] == 'hbm' else f"{f['achieved']:.0f} TFLOP/s"
        rows += f"| {names[k]} | {f['ms_alone']:.2f} | {ins(f)} | {f['launches']} | {f['bound'].upper()} | {alg} | {ach} | **{f['frac']:.2f}** | {r3[k]:.2f} ms |\n"
    else:
        rows += f"| {names[k]} | {f['ms_alone']:.2f} | {ins(f)} | {f['launches']} | - | - | - | - | {r3[k]:.2f} ms |\n"
tot = sum(f['ms_per_step'] for f in fam.values())
tin = sum((f.get('ms_in_step') or 0) for f in fam.values())
rows += f"\n(sums: {tot:.2f} ms replayed alone, one family at a time, at the capture's launch geometry; {tin:.2f} ms of kernel time inside the step, where the two streams overlap; the step: {d['ms_per_step']:.2f} ms)"
for tag, text in (('headline', head), ('families', rows)):
    s = re.sub(rf'(<!-- AUTO:{tag}:begin[^>]*-->\n).*?(\n<!-- AUTO:{tag}:end -->)', lambda m: m.group(1) + text + m.group(2), s, flags=re.S)
open(p, 'w').write(s)
print('DESIGN.md updated:', round(d['value'], 1), 'patches/s', round(d['ms_per_step'], 2), 'ms')
