#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the forward projection launches, one counter per pass (run from the repo root on the GPU box)
set -e
ROOT=$(pwd)
export TMPDIR=/tmp
rm -rf /tmp/pmc_fetch /tmp/pmc_write
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_fetch -- python3 "$ROOT/tools/pmc_linear.py" > "$ROOT/gpurun_out/pmc_fetch.log" 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_write -- python3 "$ROOT/tools/pmc_linear.py" > "$ROOT/gpurun_out/pmc_write.log" 2>&1
python3 "$ROOT/tools/pmc_linear.py" --parse /tmp/pmc_fetch /tmp/pmc_write > "$ROOT/gpurun_out/r01_pmc_linear.json"
