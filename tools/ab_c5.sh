#!/bin/bash
# A/B of library builds on the 20 000-sphere scene (config 5 at 32 samples per pixel): kernel ms per library given as argument
for lib in "$@"; do
  if [ "$lib" = default ]; then unset RT_AMD_LIB; else export RT_AMD_LIB=$PWD/$lib; fi
  timeout -k 10 300 python bench.py --no-cpu-baseline --workload c5 --spp 32 --steps 2 --warmup 1 > gpurun_out/ab5.json 2> gpurun_out/ab5.err || { echo "$lib FAILED"; tail -3 gpurun_out/ab5.err; exit 1; }
  python - "$lib" <<'PY'
import json, sys
d = json.load(open("gpurun_out/ab5.json"))
print(f"{sys.argv[1]:40s} c5 at 32 spp {d['kernel_ms_mean']:.1f} ms  checksum {d['image_checksum']}")
PY
done
