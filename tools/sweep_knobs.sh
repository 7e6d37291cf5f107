#!/bin/bash
# one box, one sitting: launch-geometry knobs of the step against the defaults (each twice, interleaved)
run() { env "$@" python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-families 2>/dev/null | python -c "import sys,json; L=sys.stdin.read().splitlines(); d=json.loads([l for l in L if l.startswith(chr(123))][0]); print('$*', round(d['ms_per_step'], 3), flush=True)"; }
for rep in 1 2; do
  run A=0
  run LTU_LA_SPLITS=256
  run LTU_LA_SPLITS=1024
  run LTU_IN_CHUNKS=1024
  run LTU_IN_CHUNKS=4096
  run LTU_WGROUP_BLOCKS=192
  run LTU_WGROUP_BLOCKS=320
  run LTU_UPW_BLOCKS=512
  run LTU_HALO_WS_BLOCKS=768
  run LTU_LA_APPLY_WAVES=8192
  run LTU_NT_RING_TNW1_BELOW=16384
  run LTU_HALO_SPLIT_BELOW=300
done
