#!/bin/bash
# A/B of library builds on the GPU box: bench.py (config 3 headline + config 1 line + camera-ray lists) once per library given as
# argument (paths relative to the repository root; "default" = raytracing_amd/librt_amd.so).  Prints ms per step, config-1 kernel ms, checksum.
for lib in "$@"; do
  if [ "$lib" = default ]; then unset RT_AMD_LIB; else export RT_AMD_LIB=$PWD/$lib; fi
  timeout -k 10 200 python bench.py --no-cpu-baseline --steps 5 > gpurun_out/ab.json 2> gpurun_out/ab.err || { echo "$lib FAILED"; tail -3 gpurun_out/ab.err; exit 1; }
  python - "$lib" <<'PY'
import json, sys
d = json.load(open("gpurun_out/ab.json"))
b = d.get("beyond_brute_force") or {}
print(f"{sys.argv[1]:40s} c3 {d['kernel_ms_mean']:.2f} ms  c1 mean {d['config1']['kernel_ms_mean']:.3f} min {d['config1']['kernel_ms_min']:.3f} ms  lists {b.get('kernel_ms_mean', 0):.2f} ms  checksum {d['image_checksum']}")
PY
done
