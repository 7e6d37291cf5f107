"""Table of issue-side counters per kernel from the two passes of tools/pmc_valu.sh.
VALU busy = SQ_ACTIVE_INST_VALU * 4 / (1024 SIMDs * dispatch cycles), dispatch cycles = GRBM_GUI_ACTIVE / 8 (summed over XCDs)."""
import csv, glob, os, re, sys
from collections import defaultdict


def collect(d):
    rows = []
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        rows += list(csv.DictReader(open(f)))
    return rows


def short(name):
    name = re.sub(r'\(.*', '', name)
    return re.sub(r'^void ', '', name)[:58]


acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(int)
for d in sys.argv[1:]:
    seen = set()
    for r in collect(d):
        k = short(r['Kernel_Name'])
        acc[k][r['Counter_Name']] += float(r['Counter_Value'])
        key = (d, r['Dispatch_Id'])
        if r['Counter_Name'] in ('GRBM_GUI_ACTIVE', 'SQ_WAVE_CYCLES') and key not in seen:
            seen.add(key)
            if r['Counter_Name'] == 'GRBM_GUI_ACTIVE':
                cnt[k] += 1
print(f'{"kernel":58s} {"n":>5s} {"us/launch":>9s} {"VALU busy":>9s} {"VALU inst/wave-cyc":>9s} {"LDS act":>8s} {"VMEM act":>8s} {"wait":>6s}')
rows = []
for k, c in acc.items():
    cyc = c.get('GRBM_GUI_ACTIVE', 0.0) / 8
    if cyc <= 0 or cnt[k] == 0:
        continue
    wc = max(c.get('SQ_WAVE_CYCLES', 0.0), 1.0)
    rows.append((cyc, k, cnt[k], cyc / cnt[k] / 2400.0, c.get('SQ_ACTIVE_INST_VALU', 0) * 4 / (1024 * cyc), c.get('SQ_INSTS_VALU', 0),
                 c.get('SQ_ACTIVE_INST_LDS', 0) / wc, c.get('SQ_ACTIVE_INST_VMEM', 0) / wc, c.get('SQ_WAIT_INST_ANY', 0) / wc))
for cyc, k, n, us, vb, iv, lds, vm, wt in sorted(rows, reverse=True)[:45]:
    print(f'{k:58s} {n:5d} {us:9.1f} {vb:9.2f} {iv / max(cyc, 1) / 1024:9.2f} {lds:8.3f} {vm:8.3f} {wt:6.2f}')
