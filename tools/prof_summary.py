"""Summarise a rocprofv3 rocpd sqlite database (the --kernel-trace --stats output of ROCm 7.2): per-kernel time per step.
usage: prof_summary.py DB STEPS [PATTERN] [--csv] [--last-steps K]
--last-steps K: EXACT per-step figures - only the kernels of the last K complete replayed steps are counted (STEPS is ignored)"""
import re, sqlite3, sys
argv = sys.argv[1:]
K = 0
if '--last-steps' in argv:
    i = argv.index('--last-steps')
    K = int(argv[i + 1])
    del argv[i:i + 2]
as_csv = '--csv' in argv
args = [a for a in argv if a != '--csv']
db = sqlite3.connect(args[0])
steps = float(args[1])
pat = args[2] if len(args) > 2 else ''
if K:
    # a step starts at the first weight_prep_chunk_kernel launch of a group; the launches behind the last start - an incomplete
    # step - are dropped
    allrows = db.execute('select name, start, end from kernels order by start').fetchall()
    st = [i for i, r in enumerate(allrows) if 'weight_prep_chunk_kernel' in r[0]]
    st = [i for k, i in enumerate(st) if k == 0 or allrows[i][1] - allrows[st[k - 1]][1] > 3000000]
    sel = allrows[st[-K - 1]:st[-1]]
    agg = {}
    for n, s0, e0 in sel:
        a = agg.setdefault(n, [0, 0, 0, 1 << 62, 0])
        a[0] += 1; a[1] += e0 - s0; a[3] = min(a[3], e0 - s0); a[4] = max(a[4], e0 - s0)
    rows = sorted(((n, a[0], a[1], a[1] / a[0], a[3], a[4]) for n, a in agg.items()), key=lambda r: -r[2])
    steps = float(K)
else:
    rows = db.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc").fetchall()
total = sum(r[2] for r in rows)
if as_csv:
    print('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"')
    for r in rows:
        print(f'"{r[0]}",{r[1]},{r[2]},{r[3]:.1f},{100.0 * r[2] / total:.3f},{r[4]},{r[5]}')
    sys.exit(0)
print(f'total {total / 1e6 / steps:.3f} ms/step over {steps:g} steps')
for r in rows:
    nm = re.sub(r'\(.*', '', r[0])[:84]
    if pat and not re.search(pat, nm):
        continue
    print(f'{r[2] / steps / 1e6:7.3f} ms {r[1] / steps:6.1f}/step {r[3] / 1e3:8.1f} us  {100.0 * r[2] / total:5.1f}%  {nm}')
