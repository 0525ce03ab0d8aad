#!/bin/bash
# Shader clock and power while (a) the render kernel, (b) the isolated filter microbenchmark run.
probe() { for k in 1 2 3 4; do rocm-smi --showclocks 2>/dev/null | grep -i -E "sclk" | head -1; rocm-smi --showpower 2>/dev/null | grep -i -E "power" | head -1; sleep 0.3; done; }
timeout -k 10 200 python bench.py --spp 4000 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/clock_bench.json 2> gpurun_out/clock_bench.err &
pid=$!
sleep 12
echo "--- render kernel"; probe
wait $pid || exit 1
( for r in 1 2 3 4 5 6 7 8 9 10 11 12; do timeout -k 10 60 ./tools/ubench_filter.bin > /dev/null || exit 1; done ) &
pid=$!
sleep 2.5
echo "--- filter microbenchmark"; probe
wait $pid
