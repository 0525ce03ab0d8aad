#!/bin/bash
# rocprofv3 kernel-trace stats of the default bench command + the bench line itself; outputs under gpurun_out/
tag=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 python bench.py > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -o p -- python bench.py --no-cpu-baseline > gpurun_out/prof_$tag.log 2>&1 || exit 1
./tools/pmc.sh $tag || exit 1
