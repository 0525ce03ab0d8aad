#!/bin/bash
# A/B of the sparse-sweep threshold (live paths per wave at or below which the spheres are spread over the lanes).
# The variant libraries are built on the CPU first (not kept in the tree): the hipcc line of raytracing_amd/csrc/Makefile with
# -DRT_FEW_PATHS=v and -o tools/librt_amd_few<v>.so.
for v in 2 4 8 16 24; do
  if [ $v = 8 ]; then unset RT_AMD_LIB; else export RT_AMD_LIB=$PWD/tools/librt_amd_few$v.so; fi
  echo "threshold $v"
  PYTHONPATH=. timeout -k 10 120 python tools/tail_probe.py | sed -n '1p;4p;5p' || exit 1
done
