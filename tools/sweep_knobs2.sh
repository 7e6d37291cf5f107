#!/bin/bash
# step-level A/B of knobs on one box (interleaved, 2 rounds)
run() { env "$@" python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-families 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith(chr(123))][0]); print('$*', round(d['ms_per_step'], 3), flush=True)"; }
for rep in 1 2; do
  run A=0
  run LTU_NT_VARIANT=1
  run LTU_NT_VARIANT=2
  run LTU_NT_SMALLTILE=2
  run LTU_NT_SMALLTILE=3
  run LTU_UPW_BLOCKS=512
  run LTU_UPW_BLOCKS=1024
done
