#!/bin/bash
# step-level A/B of knobs on one box (interleaved, 3 rounds)
run() { env "$@" python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-families 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith(chr(123))][0]); print('$*', round(d['ms_per_step'], 3), flush=True)"; }
for rep in 1 2 3; do
  run A=0
  run LTU_FUSE_ATTN_MAX_TOKENS=50000
  run LTU_FUSE_ATTN_MAX_TOKENS=10000
done
