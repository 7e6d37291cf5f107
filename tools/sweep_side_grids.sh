#!/bin/bash
# how wide should the side stream's weight-gradient kernels be?  (they run beside the main chain: narrower = politer, slower)
run() { env $1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-families 2>gpurun_out/sweep.err | python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith(chr(123))][0]); print('$1', round(d['ms_per_step'], 3), flush=True)" || tail -3 gpurun_out/sweep.err; }
for rep in 1 2 3; do
for c in "LTU_X=0" "LTU_WGROUP_BLOCKS=128,LTU_UPW_BLOCKS=128" "LTU_WGROUP_BLOCKS=96,LTU_UPW_BLOCKS=96" "LTU_WGROUP_BLOCKS=64,LTU_UPW_BLOCKS=64" "LTU_WGROUP_BLOCKS=128,LTU_UPW_BLOCKS=128,LTU_WHALO_BLOCKS=256" "LTU_WGROUP_BLOCKS=128,LTU_UPW_BLOCKS=64,LTU_WHALO_BLOCKS=192" "LTU_WGROUP_BLOCKS=96,LTU_UPW_BLOCKS=96,LTU_WHALO_BLOCKS=256"; do run "${c//,/ }"; done
done
