#!/bin/bash
# HBM-side traffic of k_trace (FETCH_SIZE, WRITE_SIZE passes, bench at 50 spp, brute-force path) per library build given as argument.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for lib in "$@"; do
  if [ "$lib" = default ]; then unset RT_AMD_LIB; else export RT_AMD_LIB=$PWD/$lib; fi
  tag=$(basename $lib .so)
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d gpurun_out/traffic_${tag}_$c -o p -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-beyond --spp 50 > gpurun_out/traffic_${tag}_$c.log 2>&1 || exit 1
  done
  python - $tag <<'PY'
import csv, glob, sys
tag = sys.argv[1]; tot = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    v = 0.0
    for f in glob.glob(f"gpurun_out/traffic_{tag}_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_trace" in r["Kernel_Name"] and r["Counter_Name"] == c: v += float(r["Counter_Value"])
    tot[c] = v * 1024
rays = 126863791.0
print(f"{tag:20s} fetch x2 {2 * tot['FETCH_SIZE'] / rays:6.1f} B/ray  write {tot['WRITE_SIZE'] / rays:6.1f} B/ray  total {(2 * tot['FETCH_SIZE'] + tot['WRITE_SIZE']) / rays:6.1f} B/ray")
PY
done
