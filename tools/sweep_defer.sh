#!/bin/bash
# one box: layers' weight-gradient groups launched together (LTU_WGRAD_DEFER_MB) against one group per layer (0), interleaved
run() { env "$@" python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-families 2>/dev/null | python -c "import sys,json; L=sys.stdin.read().splitlines(); d=json.loads([l for l in L if l.startswith(chr(123))][0]); print('$*', round(d['ms_per_step'], 3), flush=True)"; }
for rep in 1 2; do
  run LTU_WGRAD_DEFER_MB=0
  run LTU_WGRAD_DEFER_MB=128 LTU_WGROUP_NO_DIRECT=1
  run LTU_WGRAD_DEFER_MB=128
  run LTU_WGRAD_DEFER_MB=240
done
