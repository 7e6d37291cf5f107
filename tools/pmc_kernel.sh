#!/bin/bash
# issue-side and LDS counters of ONE micro-benchmark's kernels (three rocprofv3 --pmc passes): tools/pmc_kernel.sh tools/bench_upwgrad.py PATTERN
# prints per-kernel averages of every counter for kernels whose name contains PATTERN.  Run from the repo root on the GPU box.
set -e
ROOT=$(pwd)
SCRIPT=$1
PAT=$2
export TMPDIR=/tmp
i=0
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16" "SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SALU"; do
  i=$((i + 1))
  rm -rf /tmp/pmc_k$i
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d /tmp/pmc_k$i -- python3 "$ROOT/$SCRIPT" > "$ROOT/gpurun_out/pmc_k$i.log" 2>&1 || echo "pass $i failed"
done
python3 - "$PAT" <<'PY'
import csv, glob, sys, collections
pat = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('/tmp/pmc_k*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n = r.get('Kernel_Name', '')
        if pat in n:
            acc[n.split('(')[0][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
for n, cs in acc.items():
    print(n)
    for c, v in sorted(cs.items()):
        print(f'   {c:34s} {sum(v) / len(v):16.0f}   ({len(v)} launches)')
PY
