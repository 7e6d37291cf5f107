"""every kernel of a rocprofv3 --kernel-trace database as one line: start (us, relative to the first), duration, queue, workgroups,
name.  usage: trace_dump.py DB [> file]"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)").fetchall()]
qcol = 'queue_id' if 'queue_id' in cols else ('stream_id' if 'stream_id' in cols else '0')
rows = db.execute(f"select name, start, end, {qcol}, grid_x * grid_y * grid_z / (workgroup_x * workgroup_y * workgroup_z) from kernels order by start").fetchall()
t0 = rows[0][1]
for n, s, e, q, wg in rows:
    print(f'{(s - t0) / 1e3:12.1f} {(e - s) / 1e3:8.1f} q{q} {wg:6d} ' + re.sub(r'\(.*', '', n).replace('void ', '')[:70])
