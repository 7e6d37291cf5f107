#!/bin/bash
# one box, one sitting: launch-geometry knobs of the step against the defaults (each twice, interleaved); round 4: under the side stream
run() { env "$@" python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-families 2>/dev/null | python -c "import sys,json; L=sys.stdin.read().splitlines(); d=json.loads([l for l in L if l.startswith(chr(123))][0]); print('$*', round(d['ms_per_step'], 3), flush=True)"; }
for rep in 1 2; do
  run A=0
  run LTU_LA_SPLITS=256
  run LTU_LA_SPLITS=1024
  run LTU_IN_CHUNKS=512
  run LTU_IN_CHUNKS=2048
  run LTU_HALO_WS_BLOCKS=256
  run LTU_HALO_WR_BLOCKS=256
  run LTU_HALO_WR_BLOCKS=768
  run LTU_DW_BLOCKS=256
  run LTU_WHALO_BLOCKS=256
  run LTU_WGRAD_DEFER_MB=64
  run LTU_WGRAD_DEFER_MB=400
  run LTU_NO_FUSE_QKV=1
  run LTU_FUSE_QKV_MAX_TOKENS=1000000
done
