#!/bin/bash
# regenerates the files kept under profiles/ (run from the repo root on the GPU box; outputs land in gpurun_out/)
set -e
ROOT=$(pwd)
python3 bench.py > gpurun_out/r01_bench.json 2> gpurun_out/r01_bench.err
bash tools/profile_bench.sh
{
  echo "== bench_nt.py (projections: forward kernel | weight gradient incl. second stage)"; python3 tools/bench_nt.py 2>/dev/null
  echo; echo "== bench_pw.py (LayerNorm, GELU, InstanceNorm)"; python3 tools/bench_pw.py 2>/dev/null
  echo; echo "== bench_conv.py (3x3x3 convs, stride 1)"; python3 tools/bench_conv.py 2>/dev/null
  echo; echo "== bench_class.py (sub-pixel un-embedding forward)"; python3 tools/bench_class.py 2>/dev/null
  echo; echo "== bench_dwconv.py (positional depthwise conv)"; python3 tools/bench_dwconv.py 2>/dev/null
} > gpurun_out/r01_microbench_body.txt
