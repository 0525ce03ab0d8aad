#!/bin/bash
# GPU regression + bench in one call: parity tests first, bench only if they pass.
set -o pipefail
tag=${1:-run}
timeout -k 10 400 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/gpu_tests_$tag.log 2>&1
rc=$?
tail -4 gpurun_out/gpu_tests_$tag.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err
rc=$?
python - <<PY
import json
d = json.load(open("gpurun_out/bench_$tag.json"))
print("Mrays/s %.0f  ms/step %.2f  kernel_ms %.2f  frac %.3f  checksum %s" % (d["value"], d["ms_per_step"], d["kernel_ms_mean"], d["roofline"]["frac"], d["image_checksum"]))
PY
exit $rc
