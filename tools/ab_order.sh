#!/bin/bash
# A/B: tile work order (longest paths first vs image order) on C1 and C3.
set -o pipefail
for v in ordered image; do
  if [ $v = image ]; then export RT_AMD_TILE_ORDER=image; else unset RT_AMD_TILE_ORDER; fi
  timeout -k 10 120 python bench.py --workload c1 --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/ab_${v}_c1.json 2>/dev/null || exit 1
  timeout -k 10 120 python bench.py --workload c3 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/ab_${v}_c3.json 2>/dev/null || exit 1
done
python - <<'PY'
import json
for v in ("ordered", "image"):
    for w in ("c1", "c3"):
        d = json.load(open(f"gpurun_out/ab_{v}_{w}.json"))
        print(v, w, "ms/step %.3f kernel %.3f Mrays/s %.0f" % (d["ms_per_step"], d["kernel_ms_mean"], d["value"]))
PY
