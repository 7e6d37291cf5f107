#!/bin/bash
# A/B of two environment settings on ONE box, interleaved: tools/ab_env.sh "LTU_BRANCHES=0" "LTU_BRANCHES=1" [rounds]
# each run: bench.py --steps 20 --warmup 5 without the CPU baseline / family table; prints ms per step
A="$1"; B="$2"; R=${3:-3}
run() { env $1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-families 2>gpurun_out/ab_env.err | python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith(chr(123))][0]); print('$1', round(d['ms_per_step'], 3), flush=True)" || tail -5 gpurun_out/ab_env.err; }
for rep in $(seq $R); do
  run "$A"
  run "$B"
done
