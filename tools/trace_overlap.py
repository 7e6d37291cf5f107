"""Concurrency inside one replayed step of a rocprofv3 --kernel-trace database: union of the kernel intervals, time with >= 2 kernels
running, per-queue busy time, and (with --list) the launch list with queue ids.  usage: trace_overlap.py DB [--list] [--which K]"""
import re, sqlite3, sys
args = [a for a in sys.argv[1:] if not a.startswith('--')]
db = sqlite3.connect(args[0])
which = int(sys.argv[sys.argv.index('--which') + 1]) if '--which' in sys.argv else 2
cols = [r[1] for r in db.execute("pragma table_info(kernels)").fetchall()]
if '--cols' in sys.argv:
    print(cols)
qcol = 'queue_id' if 'queue_id' in cols else ('stream_id' if 'stream_id' in cols else None)
# grid / workgroup / LDS columns where the database has them (--list prints them: which side kernels leave workgroups pending?)
extra = {}
if '--list' in sys.argv and all(c in cols for c in ('grid_x', 'grid_y', 'grid_z', 'workgroup_x', 'workgroup_y', 'workgroup_z', 'lds_size')):
    vg = ', vgpr_count, accum_vgpr_count' if 'vgpr_count' in cols and 'accum_vgpr_count' in cols else ', 0, 0'
    for r in db.execute("select start, grid_x * grid_y * grid_z / (workgroup_x * workgroup_y * workgroup_z), workgroup_x * workgroup_y * workgroup_z, lds_size" + vg + " from kernels").fetchall():
        extra[r[0]] = ('%d wg x %d thr, lds %d, vgpr %d+%d' % (r[1], r[2], r[3], r[4], r[5]),)
rows = db.execute(f"select name, start, end, {qcol or '0'} from kernels order by start").fetchall()
starts = [i for i, r in enumerate(rows) if 'weight_prep_chunk_kernel' in r[0]]
# a step refreshes its weight operands in up to three launches (encoder / decoder / bridges): the first of each group starts the step
starts = [i for k, i in enumerate(starts) if k == 0 or rows[i][1] - rows[starts[k - 1]][1] > 3000000]
i0, i1 = starts[-which - 1], starts[-which]
step = rows[i0:i1]
short = lambda n: re.sub(r'\(.*', '', n).replace('void ', '')[:60]
t0 = step[0][1]
ev = []
for n, s, e, q in step:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
cur, last, union, multi = 0, None, 0, 0
for t, d in ev:
    if last is not None and cur > 0:
        union += t - last
        if cur > 1:
            multi += t - last
    cur += d; last = t
span = max(e for _, _, e, _ in step) - t0
busy = sum(e - s for _, s, e, _ in step)
print(f'{len(step)} launches, span {span/1e6:.3f} ms, sum of durations {busy/1e6:.3f} ms, union {union/1e6:.3f} ms, >=2 kernels running {multi/1e6:.3f} ms')
perq = {}
for n, s, e, q in step:
    a = perq.setdefault(q, [0, 0]); a[0] += e - s; a[1] += 1
for q, (t, c) in sorted(perq.items(), key=lambda kv: -kv[1][0]):
    print(f'  queue {q}: {t/1e6:.3f} ms busy, {c} launches')
if '--list' in sys.argv:
    for i, (n, s, e, q) in enumerate(step):
        print(f'{i:4d} q{q} {(s - t0)/1e3:9.1f} .. {(e - t0)/1e3:9.1f} us  {(e - s)/1e3:7.1f} us  {short(n)}' + ('   [' + ' '.join(str(v) for v in extra[s]) + ']' if s in extra else ''))
