#!/bin/bash
# Segment / chunk size of the work hand-out (RT_AMD_CHUNK_FACTOR x sqrt(items per resident wave)) on the camera-ray-list path,
# megakernel and hierarchy walk, config 3 and config 1.
for f in 1.0 1.5 2.0 3.0; do
  for m in megakernel bvh; do
    for w in c3 c1; do
      RT_AMD_CHUNK_FACTOR=$f timeout -k 10 200 python bench.py --workload $w --mode $m --steps 3 --warmup 1 --no-cpu-baseline --no-beyond --no-config1 --camera-ray-lists > gpurun_out/cf.json 2> gpurun_out/cf.err || exit 1
      python -c "
import json; d = json.load(open('gpurun_out/cf.json'))
print('factor $f  %-10s %s  kernel %8.3f ms' % ('$m', '$w', d['kernel_ms_mean']))"
    done
  done
done
