#!/bin/bash
# one box, one sitting: the default bench under runtime environment variables that change how HIP graphs dispatch kernels
run() { env "$@" python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; L=sys.stdin.read().splitlines(); d=json.loads([l for l in L if l.startswith(chr(123))][0]); print('$*', round(d['ms_per_step'], 3), flush=True)"; }
for rep in 1 2; do
  run A=0
  run DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
  run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
  run HIP_FORCE_DEV_KERNARG=1
  run HIP_FORCE_DEV_KERNARG=0
  run GPU_MAX_HW_QUEUES=1
  run GPU_MAX_HW_QUEUES=2
  run GPU_MAX_HW_QUEUES=8
  run HSA_NO_SCRATCH_RECLAIM=1
  run DEBUG_HIP_GRAPH_DOT_PRINT=0 AMD_DIRECT_DISPATCH=1
done
