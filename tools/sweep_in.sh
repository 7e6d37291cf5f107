#!/bin/bash
# InstanceNorm microbench under the vector-width / fold / chunk knobs
for env in "A=0" "LTU_IN_NO_FOLD=1" "LTU_IN_FOLD_BLOCKS=512" "LTU_IN_FOLD_BLOCKS=1024" "LTU_IN_VW8=1"; do
  echo "== $env"
  env $env python tools/bench_pw.py in 2>/dev/null | tail -4
done
