"""Where a kernel sits in the launch order of the last step of a rocprofv3 kernel trace (rocpd sqlite): its neighbours.
usage: trace_neighbours.py DB PATTERN [count]"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
pat = sys.argv[2]
cnt = int(sys.argv[3]) if len(sys.argv) > 3 else 40
rows = db.execute("select name, start, end from kernels order by start").fetchall()
short = lambda n: re.sub(r'\(.*', '', n)[:70]
idx = [i for i, r in enumerate(rows) if re.search(pat, r[0])]
print(len(rows), 'kernels,', len(idx), 'match')
for i in idx[-cnt:]:
    print(i, f'{(rows[i][2] - rows[i][1]) / 1e3:6.1f} us | prev: {short(rows[i - 1][0])} | next: {short(rows[i + 1][0]) if i + 1 < len(rows) else "-"}')
