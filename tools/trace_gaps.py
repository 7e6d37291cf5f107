"""idle stretches of the main queue in a launch list written by tools/trace_overlap.py --list (tools/trace_env.sh):
usage: trace_gaps.py gpurun_out/X_overlap.txt [min_us]"""
import re, sys
rows = []
for l in open(sys.argv[1]):
    m = re.match(r'\s*(\d+) q(\d)\s+([\d.]+) \.\.\s+([\d.]+) us\s+([\d.]+) us\s+(.*)', l)
    if m:
        rows.append((int(m[1]), int(m[2]), float(m[3]), float(m[4]), float(m[5]), m[6].strip()))
mq = min(r[1] for r in rows)
main = [r for r in rows if r[1] == mq and 'rocclr' not in r[5] and 'at::native' not in r[5]]
lim = float(sys.argv[2]) if len(sys.argv) > 2 else 10.0
gaps = [(main[i + 1][2] - main[i][3], main[i], main[i + 1]) for i in range(len(main) - 1)]
print('main queue: %d launches, busy %.3f ms, span %.3f ms, idle %.3f ms in %d gaps > 0.5 us' % (
    len(main), sum(r[4] for r in main) / 1e3, (main[-1][3] - main[0][2]) / 1e3, sum(g[0] for g in gaps if g[0] > 0.5) / 1e3,
    sum(1 for g in gaps if g[0] > 0.5)))
for g, a, b in gaps:
    if g >= lim:
        side = [r for r in rows if r[1] != mq and r[3] > a[3] and r[2] < b[2]]
        print('%7.1f us idle at %8.1f us: after #%d %s, before #%d %s; side meanwhile: %s' % (
            g, a[3], a[0], a[5][:34], b[0], b[5][:34], ', '.join('%s %.0f' % (re.sub(r'_bf16_kernel|_kernel|<.*', '', r[5])[:18], r[4]) for r in side[:12])))
