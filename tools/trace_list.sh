#!/bin/bash
# one replayed step as a launch list with timestamps (gpurun_out/step_list.txt)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_list
rocprofv3 --kernel-trace -d /tmp/prof_list -o t -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-families --steps 6 --warmup 3 > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/trace_list.err
DB=$(find /tmp/prof_list -name '*.db' | head -1)
python3 $GRAFT_REPO_ROOT/tools/trace_step.py "$DB" --list > $GRAFT_REPO_ROOT/gpurun_out/step_list.txt
