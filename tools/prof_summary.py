"""Summarise a rocprofv3 rocpd sqlite database: per-kernel time per step.  usage: prof_summary.py DB STEPS [PATTERN]"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
steps = float(sys.argv[2])
pat = sys.argv[3] if len(sys.argv) > 3 else ''
rows = db.execute("select name, count(*), sum(end-start), avg(end-start) from kernels group by name order by 3 desc").fetchall()
print(f'total {sum(r[2] for r in rows) / 1e6 / steps:.3f} ms/step')
for r in rows:
    nm = re.sub(r'\(.*', '', r[0])[:84]
    if pat and not re.search(pat, nm):
        continue
    print(f'{r[2] / steps / 1e6:7.3f} ms {r[1] / steps:6.1f}/step {r[3] / 1e3:8.1f} us  {nm}')
