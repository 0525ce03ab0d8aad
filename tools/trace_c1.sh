#!/bin/bash
# Kernel durations of a config-1 step (rocprofv3 kernel trace).
set -o pipefail
cd /tmp && export TMPDIR=/tmp
timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_c1 -- python $GRAFT_REPO_ROOT/bench.py --workload c1 --steps 10 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_c1.log 2>&1 || exit 1
cd $GRAFT_REPO_ROOT && find gpurun_out/prof_c1 -name "*kernel_stats.csv" | head -1 | xargs cut -c1-150 | head -8
