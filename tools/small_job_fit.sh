#!/bin/bash
# Fixed cost of a render call: step time of the 1200x800 final scene at increasing sample counts.
set -o pipefail
for s in 1 2 5 10 20 40 80; do
  timeout -k 10 120 python bench.py --workload c1 --spp $s --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/fit_$s.json 2> gpurun_out/fit_$s.err || exit 1
done
python - <<'PY'
import json
for s in (1, 2, 5, 10, 20, 40, 80):
    d = json.load(open(f"gpurun_out/fit_{s}.json"))
    print("spp %3d  ms/step %.3f  kernel_ms %.3f  Mrays/s %.0f" % (s, d["ms_per_step"], d["kernel_ms_mean"], d["value"]))
PY
cd /tmp && export TMPDIR=/tmp
timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_c1 -- python $GRAFT_REPO_ROOT/bench.py --workload c1 --steps 10 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_c1.log 2>&1 || exit 1
cd $GRAFT_REPO_ROOT && find gpurun_out/prof_c1 -name "*kernel_stats.csv" | head -1 | xargs cat
