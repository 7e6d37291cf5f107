#!/bin/bash
# hardware counters of the bench step per kernel family (three rocprofv3 --pmc passes); run from the repo root on the GPU box
set -e
ROOT=$(pwd)
export TMPDIR=/tmp
rm -rf /tmp/pmc_a /tmp/pmc_b /tmp/pmc_c
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_a -- python3 "$ROOT/tools/pmc_step.py" > "$ROOT/gpurun_out/pmc_a.log" 2>&1
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_b -- python3 "$ROOT/tools/pmc_step.py" > "$ROOT/gpurun_out/pmc_b.log" 2>&1
timeout -k 10 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d /tmp/pmc_c -- python3 "$ROOT/tools/pmc_step.py" > "$ROOT/gpurun_out/pmc_c.log" 2>&1
python3 "$ROOT/tools/pmc_step.py" --parse /tmp/pmc_a /tmp/pmc_b /tmp/pmc_c > "$ROOT/gpurun_out/${1:-r03}_pmc_step.json"
