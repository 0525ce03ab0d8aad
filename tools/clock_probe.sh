#!/bin/bash
# Shader clock while the render kernel runs: rocm-smi samples next to a long bench run.
timeout -k 10 200 python bench.py --spp 4000 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/clock_bench.json 2> gpurun_out/clock_bench.err &
pid=$!
sleep 12
for k in 1 2 3 4 5 6; do
  rocm-smi --showclocks 2>/dev/null | grep -i -E "sclk|fclk|mclk" | head -4
  rocm-smi --showpower 2>/dev/null | grep -i -E "power" | head -2
  sleep 0.4
done
wait $pid
python -c "
import json; d=json.load(open('gpurun_out/clock_bench.json')); print('ms/step', d['ms_per_step'], 'Mrays/s', d['value'])"
