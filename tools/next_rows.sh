#!/bin/bash
# Measurements of the steps around the render loop: PPM text encode (GPU and host) and the main_modern drop-in.
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/next_c3.json 2> gpurun_out/next_c3.err || exit 1
timeout -k 10 200 python bench.py --workload c1 --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/next_c1.json 2> gpurun_out/next_c1.err || exit 1
timeout -k 10 300 python bench.py --workload c4 --spp 64 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/next_c4.json 2> gpurun_out/next_c4.err || exit 1
timeout -k 10 120 bin/raytracing_modern_amd 2560 1440 10 /tmp/modern_bands.ppm 0 64 > gpurun_out/next_modern_bands.txt 2>&1 || exit 1
timeout -k 10 120 bin/raytracing_modern_amd 2560 1440 10 /tmp/modern_one.ppm 0 0 > gpurun_out/next_modern_one.txt 2>&1 || exit 1
cmp /tmp/modern_bands.ppm /tmp/modern_one.ppm && echo "band-by-band and single-call output.ppm identical" >> gpurun_out/next_modern_one.txt
python - <<'PY'
import json
for n in ("c1", "c3", "c4"):
    d = json.load(open(f"gpurun_out/next_{n}.json"))
    print(n, "ms/step %.2f" % d["ms_per_step"], d["ppm_encode"])
PY
cat gpurun_out/next_modern_bands.txt gpurun_out/next_modern_one.txt
