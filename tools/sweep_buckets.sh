#!/bin/bash
# one box, one sitting: bench.py --rehearse-comm over all-reduce modes and bucket plans, each twice, interleaved (box-to-box spread is +-2 %)
run() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-families "$@" 2>/dev/null | python -c "import sys,json; L=sys.stdin.read().splitlines(); d=json.loads([l for l in L if l.startswith(chr(123))][0]); print('$*', round(d['ms_per_step'], 3), flush=True)"; }
for rep in 1 2; do
  run
  for mode in segments graph; do
    for plan in "100 0" "48 0" "32 0" "32 0.5" "16 0" "8 0"; do set -- $plan; run --rehearse-comm --allreduce $mode --bucket-mb $1 --tail-mb $2; done
  done
  run --rehearse-comm --allreduce after
done
