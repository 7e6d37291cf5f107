#!/bin/bash
# weight-gradient queue: batch size and side-stream priority, interleaved on one box (ms per step)
run() { env $1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-families 2>gpurun_out/sweep_wq.err | python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith(chr(123))][0]); print('$1', round(d['ms_per_step'], 3), flush=True)" || tail -3 gpurun_out/sweep_wq.err; }
python -c "import torch; print('priority range', torch.cuda.Stream.priority_range())"
for rep in 1 2; do
  for cfg in "LTU_WQ=0" "LTU_WQ=1" "LTU_WQ_JOBS=8" "LTU_WQ_JOBS=4" "LTU_WQ_JOBS=2" "LTU_WQ_PRIO=1" "LTU_WQ_PRIO=-1" "LTU_WQ_JOBS=4,LTU_WQ_PRIO=1"; do
    run "${cfg//,/ }"
  done
done
