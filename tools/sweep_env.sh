#!/bin/bash
# usage: sweep_env.sh VAR v1 v2 ... -- PATTERN   : rocprof kernel-time total of the bench step and of kernels matching PATTERN per value
VAR=$1; shift
VALS=()
while [ "$1" != "--" ]; do VALS+=("$1"); shift; done
shift
PAT=$1
for v in "${VALS[@]}"; do
  export $VAR=$v
  bash tools/profile_bench.sh > /dev/null 2>&1
  echo "$VAR=$v: $(head -1 gpurun_out/prof_bench_stats.txt | cut -d' ' -f1-3)  match=$(grep -E "$PAT" gpurun_out/prof_bench_stats.txt | awk '{s+=$1} END {print s}') ms"
done
