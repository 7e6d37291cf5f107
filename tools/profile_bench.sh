#!/bin/bash
# rocprofv3 kernel trace of the default bench run; leaves the per-kernel summary under gpurun_out/ (run from the repo root on the GPU box)
set -e
ROOT=$(pwd)
export TMPDIR=/tmp
rm -rf /tmp/prof_bench
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_bench -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-families --steps 10 --warmup 3 > "$ROOT/gpurun_out/prof_bench.log" 2>&1
DB=$(find /tmp/prof_bench -name '*.db' | head -1)
python3 "$ROOT/tools/prof_summary.py" "$DB" 8 --last-steps 8 > "$ROOT/gpurun_out/prof_bench_stats.txt"
python3 "$ROOT/tools/prof_summary.py" "$DB" 8 --last-steps 8 --csv > "$ROOT/gpurun_out/prof_bench_stats.csv"
tail -1 "$ROOT/gpurun_out/prof_bench.log" | cut -c1-200
