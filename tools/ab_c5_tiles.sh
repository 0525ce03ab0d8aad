#!/bin/bash
# The 20 000-sphere scene (config 5 at 32 samples per pixel) with the filter's operands as LDS tiles shared by a block
# (RT_AMD_LARGE_SCENES=tiles) and with every wave streaming them from L2 (the library's default), brute force and with camera-ray
# lists; kernel ms and checksum
for rep in 1 2; do for mode in tiles stream; do for lists in "" "--camera-ray-lists"; do
  export RT_AMD_LARGE_SCENES=$mode
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-beyond --workload c5 --spp 32 --steps 3 --warmup 1 $lists > gpurun_out/ab5.json 2> gpurun_out/ab5.err || { echo "$mode FAILED"; tail -3 gpurun_out/ab5.err; exit 1; }
  python - "$mode $lists" <<'PY'
import json, sys
d = json.load(open("gpurun_out/ab5.json"))
print(f"{sys.argv[1]:32s} c5 at 32 spp {d['kernel_ms_mean']:.1f} ms  checksum {d['image_checksum']}")
PY
done; done; done
