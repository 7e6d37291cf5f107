#!/bin/bash
# rocprofv3 kernel trace of an arbitrary python tool: tools/prof_cmd.sh <divisor> tools/x.py [args]; summary -> gpurun_out/prof_cmd.txt
set -e
ROOT=$(pwd)
export TMPDIR=/tmp
DIV=$1; shift
rm -rf /tmp/prof_cmd
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_cmd -- python3 "$@" > "$ROOT/gpurun_out/prof_cmd.log" 2>&1
DB=$(find /tmp/prof_cmd -name '*.db' | head -1)
python3 "$ROOT/tools/prof_summary.py" "$DB" "$DIV" > "$ROOT/gpurun_out/prof_cmd.txt"
