#!/bin/bash
# A/B of the two filters in one box: parity tests under each, then bench.
for f in plane disc; do
  export RT_AMD_FILTER=$f
  timeout -k 10 400 python -u -m pytest tests -m gpu -x -q --timeout 200 --timeout-method thread > gpurun_out/gpu_tests_$f.log 2>&1 || { tail -20 gpurun_out/gpu_tests_$f.log; exit 1; }
  tail -1 gpurun_out/gpu_tests_$f.log
  for w in c3 c5; do
    extra=""; [ $w = c5 ] && extra="--spp 8"
    timeout -k 10 300 python bench.py --workload $w $extra --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_${f}_$w.json 2> gpurun_out/bench_${f}_$w.err || exit 1
    python -c "
import json; d = json.load(open('gpurun_out/bench_${f}_$w.json'))
print('$f $w Mrays/s %.0f ms/step %.2f frac %.3f checksum %s' % (d['value'], d['ms_per_step'], d['roofline']['frac'], d['image_checksum']))"
  done
done
