#!/bin/bash
# weight-gradient group size (operand MB launched together) under the side stream, interleaved on one box
run() { env "$@" python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-families 2>/dev/null | python -c "import sys,json; L=sys.stdin.read().splitlines(); d=json.loads([l for l in L if l.startswith(chr(123))][0]); print('$*', round(d['ms_per_step'], 3), flush=True)"; }
for rep in 1 2 3; do
  for mb in 128 256 400 800 1600 4000; do run LTU_WGRAD_DEFER_MB=$mb; done
done
