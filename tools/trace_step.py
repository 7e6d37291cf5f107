"""One replayed step out of a rocprofv3 --kernel-trace database: per-kernel sums, busy time vs span (idle gaps between consecutive
dispatches), and the launch list with timestamps.  A step starts at weight_prep_chunk_kernel (first launch of every step).
usage: trace_step.py DB [--list] [--which K]   (K: the K-th step from the end, default 2 = a graph replay in the timed region)"""
import re, sqlite3, sys
args = [a for a in sys.argv[1:] if not a.startswith('--')]
db = sqlite3.connect(args[0])
which = int(sys.argv[sys.argv.index('--which') + 1]) if '--which' in sys.argv else 2
rows = db.execute('select name, start, end from kernels order by start').fetchall()
starts = [i for i, r in enumerate(rows) if 'weight_prep_chunk_kernel' in r[0]]
# a step refreshes its weight operands in up to three launches (encoder / decoder / bridges): the first of each group starts the step
starts = [i for k, i in enumerate(starts) if k == 0 or rows[i][1] - rows[starts[k - 1]][1] > 3000000]
i0, i1 = starts[-which - 1], starts[-which]
step = rows[i0:i1]
short = lambda n: re.sub(r'\(.*', '', n).replace('void ', '')[:70]
span = step[-1][2] - step[0][1]
busy = sum(e - s for _, s, e in step)
gaps = [step[i + 1][1] - step[i][2] for i in range(len(step) - 1)]
pos = sum(g for g in gaps if g > 0)
print(f'{len(step)} launches, span {span / 1e6:.3f} ms, sum of kernel durations {busy / 1e6:.3f} ms, idle between kernels {pos / 1e6:.3f} ms '
      f'(overlap {-sum(g for g in gaps if g < 0) / 1e6:.3f} ms); median gap {sorted(gaps)[len(gaps) // 2] / 1e3:.2f} us')
agg = {}
for n, s, e in step:
    a = agg.setdefault(short(n), [0, 0])
    a[0] += e - s
    a[1] += 1
for n, (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    print(f'{t / 1e6:7.3f} ms {c:4d} x {t / c / 1e3:7.1f} us  {100.0 * t / busy:5.1f}%  {n}')
if '--list' in sys.argv:
    t0 = step[0][1]
    for i, (n, s, e) in enumerate(step):
        print(f'{i:4d} {(s - t0) / 1e3:9.1f} us  {(e - s) / 1e3:7.1f} us  gap {gaps[i - 1] / 1e3 if i else 0:6.2f}  {short(n)}')
