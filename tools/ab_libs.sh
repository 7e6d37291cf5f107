#!/bin/bash
# A/B of two builds of libltu_hip.so on one box: lintransunet_amd/libltu_old.bin against libltu_new.bin, interleaved
run() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-families 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith(chr(123))][0]); print('$1', round(d['ms_per_step'], 3), flush=True)"; }
for rep in 1 2 3; do
  cp lintransunet_amd/libltu_old.bin lintransunet_amd/libltu_hip.so; run old
  cp lintransunet_amd/libltu_new.bin lintransunet_amd/libltu_hip.so; run new
done
