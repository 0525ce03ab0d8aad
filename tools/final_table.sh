#!/bin/bash
# One line per workload and kernel organisation with the shipped library (for profiles/README.md).
set -o pipefail
run() { # workload mode steps
  timeout -k 10 600 python bench.py --workload $1 --mode $2 --steps $3 --warmup 1 --no-cpu-baseline > gpurun_out/final_$1_$2.json 2> gpurun_out/final_$1_$2.err || exit 1
  python -c "
import json; d = json.load(open('gpurun_out/final_$1_$2.json'))
print('%-3s %-10s ms/step %9.2f  Mrays/s %8.0f  Msamples/s %7.0f  rays/sample %.3f  checksum %d' % ('$1', '$2', d['ms_per_step'], d['value'], d['msamples_per_s'], d['rays_per_sample'], d['image_checksum']))"
}
run c1 megakernel 10; run c1 bvh 10; run c1 wavefront 3; run c1 pixel 3
run c2 megakernel 10; run c2 bvh 10
run c3 megakernel 3; run c3 bvh 3
run c4 megakernel 1; run c4 bvh 1
run c5 megakernel 1; run c5 bvh 2
