"""Summarise a rocprofv3 rocpd sqlite database (the --kernel-trace --stats output of ROCm 7.2): per-kernel time per step.
usage: prof_summary.py DB STEPS [PATTERN] [--csv]"""
import re, sqlite3, sys
args = [a for a in sys.argv[1:] if a != '--csv']
as_csv = '--csv' in sys.argv
db = sqlite3.connect(args[0])
steps = float(args[1])
pat = args[2] if len(args) > 2 else ''
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
