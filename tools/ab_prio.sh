#!/bin/bash
# A/B of the depth thresholds for wave issue priority: default 8/16/32, 12/20/32, 16/24/36, 4/8/16, none.
# The variant libraries are built on the CPU first (not kept in the tree): the hipcc line of raytracing_amd/csrc/Makefile with
# -DRT_PRIO_1=a -DRT_PRIO_2=b -DRT_PRIO_3=c and -o tools/librt_amd_prio<k>.so.
for v in 0 1 2 3 4; do
  if [ $v = 0 ]; then unset RT_AMD_LIB; else export RT_AMD_LIB=$PWD/tools/librt_amd_prio$v.so; fi
  echo "variant $v"
  PYTHONPATH=. timeout -k 10 120 python tools/tail_probe.py | sed -n '4p;5p' || exit 1
  timeout -k 10 120 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c3 kernel %.2f ms' % d['kernel_ms_mean'])" || exit 1
done
