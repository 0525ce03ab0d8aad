#!/bin/bash
# Shader clock and power while the render kernel runs: rocm-smi samples next to a long bench run (about 15 s of kernels).
timeout -k 10 200 python bench.py --spp 20000 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/clock_bench.json 2> gpurun_out/clock_bench.err &
pid=$!
sleep 9
for k in 1 2 3 4 5 6; do
  rocm-smi --showclocks 2>/dev/null | grep -i -E "sclk" | head -1
  rocm-smi --showpower 2>/dev/null | grep -i -E "Power \(W\)|Socket" | head -1
  sleep 0.7
done
wait $pid
python -c "
import json; d=json.load(open('gpurun_out/clock_bench.json')); print('ms/step', d['ms_per_step'], 'Mrays/s', d['value'])"
