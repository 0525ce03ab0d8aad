#!/bin/bash
# k_primary's own duration per library build given as argument (rocprofv3 kernel trace of the config-3 command, 3 steps), after the
# rate at which this GPU fills 7.7 GB with a plain store kernel (what a kernel that only writes the queue could reach).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 120 python - <<'PY' || exit 1
import torch
x = torch.empty(240_000_000 * 8, dtype=torch.float32, device="cuda")
for _ in range(2): x.fill_(1.0)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): x.fill_(2.0)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
print(f"fill of {x.numel() * 4 / 1e9:.2f} GB: {ms:.3f} ms = {x.numel() * 4 / ms / 1e9:.2f} TB/s")
PY
for lib in "$@"; do
  if [ "$lib" = default ]; then unset RT_AMD_LIB; else export RT_AMD_LIB=$PWD/$lib; fi
  name=$(basename $lib .so)
  rm -rf gpurun_out/pr_$name
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pr_$name -o p -- python bench.py --no-cpu-baseline --no-beyond --no-config1 --steps 3 > gpurun_out/pr_$name.log 2>&1 || { echo "$lib FAILED"; exit 1; }
  python - "$lib" gpurun_out/pr_$name <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[2] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if r["Name"].startswith(("k_primary", "k_trace", "k_resolve")):
        print(f"{sys.argv[1]:28s} {r['Name'][:40]:40s} mean {float(r['AverageNs']) / 1e6:.3f} ms  min {float(r['MinNs']) / 1e6:.3f}")
PY
done
