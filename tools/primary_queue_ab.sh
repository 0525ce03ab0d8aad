#!/bin/bash
# Camera rays by a kernel of their own (k_primary, the default when camera rays are swept) against camera rays staged inside k_trace
# (RT_AMD_PRIMARY_KERNEL=0): config 3 and config 1 on the brute-force path, whole step (HIP events around all kernels of the render
# call) and per kernel (rocprofv3 --kernel-trace --stats).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for mode in 1 0; do
  export RT_AMD_PRIMARY_KERNEL=$mode
  for w in c3 c1; do
    timeout -k 10 200 python bench.py --no-cpu-baseline --no-beyond --no-config1 --workload $w --steps 5 > gpurun_out/pq.json 2> gpurun_out/pq.err || { echo "$mode $w FAILED"; tail -3 gpurun_out/pq.err; exit 1; }
    python - $mode $w <<'PY'
import json, sys
d = json.load(open("gpurun_out/pq.json"))
print(f"RT_AMD_PRIMARY_KERNEL={sys.argv[1]} {sys.argv[2]}: all kernels of a step {d['kernel_ms_mean']:.3f} ms, wall {d['ms_per_step']:.3f} ms, checksum {d['image_checksum']}")
PY
  done
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pq_$mode -o p -- python bench.py --no-cpu-baseline --no-beyond --no-config1 --steps 5 > gpurun_out/pq_$mode.log 2>&1 || exit 1
  python - $mode <<'PY'
import csv, glob, sys
for f in glob.glob(f"gpurun_out/pq_{sys.argv[1]}/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Name"].startswith(("void k_trace", "k_primary", "k_resolve")):
            print(f"    {r['Name'][:56]:56s} calls {r['Calls']:>3s}  mean {float(r['AverageNs']) / 1e6:8.3f} ms")
PY
done
