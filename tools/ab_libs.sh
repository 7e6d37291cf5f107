#!/bin/bash
# A/B of two builds of libltu_hip.so on ONE box (the boxes of the pool differ by up to 4 %): lintransunet_amd/libltu_old.bin against
# libltu_new.bin, three interleaved rounds of `bench.py --steps 20 --warmup 5`.  Preparing the two files (in this container):
#   mkdir -p /tmp/oldsrc && git archive <base commit> lintransunet_amd/csrc include | tar -x -C /tmp/oldsrc
#   make -C /tmp/oldsrc/lintransunet_amd/csrc -j8 && cp /tmp/oldsrc/lintransunet_amd/libltu_hip.so lintransunet_amd/libltu_old.bin
#   make -C lintransunet_amd/csrc -j8 && cp lintransunet_amd/libltu_hip.so lintransunet_amd/libltu_new.bin
#   gpurun -- 'bash tools/ab_libs.sh'        (*.bin is git-ignored and travels with the snapshot; delete both afterwards)
run() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-families 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith(chr(123))][0]); print('$1', round(d['ms_per_step'], 3), flush=True)"; }
for rep in 1 2 3; do
  cp lintransunet_amd/libltu_old.bin lintransunet_amd/libltu_hip.so; run old
  cp lintransunet_amd/libltu_new.bin lintransunet_amd/libltu_hip.so; run new
done
