#!/bin/bash
# Issue-side counters of the bench step per kernel (two rocprofv3 --pmc passes): which kernels are VALU-issue-bound rather than
# memory-bound.  Run from the repo root on the GPU box; prints a table (tools/pmc_valu.py).
set -e
ROOT=$(pwd)
export TMPDIR=/tmp
rm -rf /tmp/pmc_v1 /tmp/pmc_v2
timeout -k 10 400 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_v1 -- python3 "$ROOT/tools/pmc_step.py" > "$ROOT/gpurun_out/pmc_v1.log" 2>&1
timeout -k 10 400 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d /tmp/pmc_v2 -- python3 "$ROOT/tools/pmc_step.py" > "$ROOT/gpurun_out/pmc_v2.log" 2>&1
python3 "$ROOT/tools/pmc_valu.py" /tmp/pmc_v1 /tmp/pmc_v2 > "$ROOT/gpurun_out/pmc_valu.txt"
