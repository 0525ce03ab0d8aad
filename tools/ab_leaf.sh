#!/bin/bash
# A/B of the leaf size of the sphere hierarchy (RT_MODE_BVH) on configs 5 and 3.
# The variant libraries are built on the CPU first (not kept in the tree): the hipcc line of raytracing_amd/csrc/Makefile with
# -DRT_BVH_LEAF=v and -o tools/librt_amd_leaf<v>.so.
for v in 2 4 8; do
  if [ $v = 4 ]; then unset RT_AMD_LIB; else export RT_AMD_LIB=$PWD/tools/librt_amd_leaf$v.so; fi
  for w in c5 c3; do
    timeout -k 10 300 python bench.py --workload $w --mode bvh --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/leaf_${v}_$w.json 2>/dev/null || exit 1
    python -c "
import json; d=json.load(open('gpurun_out/leaf_${v}_$w.json')); print('leaf $v', '$w', '%.1f ms' % d['ms_per_step'], d['image_checksum'])"
  done
done
