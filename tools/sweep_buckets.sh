#!/bin/bash
# one box, one sitting: bench.py --rehearse-comm over bucket plans, each twice, interleaved (box-to-box spread is +-2 %)
run() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; L=sys.stdin.read().splitlines(); d=json.loads([l for l in L if l.startswith(chr(123))][0]); print('$*', round(d['ms_per_step'], 3), flush=True)"; }
for rep in 1 2; do
  run
  for plan in "100 0" "48 0" "48 0.5" "32 0" "32 0.5" "16 0" "16 0.5" "8 0"; do set -- $plan; run --rehearse-comm --bucket-mb $1 --tail-mb $2; done
  run --rehearse-comm --allreduce after
done
