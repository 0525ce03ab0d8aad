#!/bin/bash
# One line per workload and kernel organisation with the shipped library (for profiles/README.md).
set -o pipefail
run() { # workload mode steps [extra bench flags]   (megakernel without flags = the brute-force path bench.py times; "--camera-ray-lists" = the library's default)
  w=$1; m=$2; st=$3; shift 3
  timeout -k 10 600 python bench.py --workload $w --mode $m --steps $st --warmup 1 --no-cpu-baseline --no-beyond --no-config1 "$@" > gpurun_out/final_${w}_$m.json 2> gpurun_out/final_${w}_$m.err || exit 1
  python -c "
import json; d = json.load(open('gpurun_out/final_${w}_$m.json'))
print('%-3s %-10s %-20s ms/step %9.2f  Mrays/s %8.0f  Msamples/s %7.0f  rays/sample %.3f  checksum %d' % ('$w', '$m', '$*', d['ms_per_step'], d['value'], d['msamples_per_s'], d['rays_per_sample'], d['image_checksum']))"
}
run c1 megakernel 10; run c1 megakernel 10 --camera-ray-lists; run c1 bvh 10 --camera-ray-lists; run c1 wavefront 3; run c1 pixel 3
run c2 megakernel 10; run c2 megakernel 10 --camera-ray-lists; run c2 bvh 10 --camera-ray-lists
run c3 megakernel 3; run c3 megakernel 3 --camera-ray-lists; run c3 bvh 3 --camera-ray-lists; run c3 wavefront 2
run c4 megakernel 1; run c4 megakernel 1 --camera-ray-lists
run c5 megakernel 1; run c5 megakernel 1 --camera-ray-lists; run c5 bvh 2 --camera-ray-lists
