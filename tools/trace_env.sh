#!/bin/bash
# kernel trace of bench.py under an environment setting: tools/trace_env.sh "LTU_BRANCHES=1" out_prefix
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_env
env $1 rocprofv3 --kernel-trace -d /tmp/prof_env -o t -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-families --steps 6 --warmup 3 > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/$2.err
DB=$(find /tmp/prof_env -name '*.db' | head -1)
python3 $GRAFT_REPO_ROOT/tools/trace_overlap.py "$DB" --list --cols > $GRAFT_REPO_ROOT/gpurun_out/$2_overlap.txt
python3 $GRAFT_REPO_ROOT/tools/trace_step.py "$DB" > $GRAFT_REPO_ROOT/gpurun_out/$2_step.txt
