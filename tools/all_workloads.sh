#!/bin/bash
# One bench line per workload (c1, c2, c3, c5) for the record.
for w in c1 c2 c3 c5; do
  steps=5; [ $w = c5 ] && steps=2
  timeout -k 10 400 python bench.py --workload $w --steps $steps --warmup 1 --no-cpu-baseline > gpurun_out/bench_all_$w.json 2> gpurun_out/bench_all_$w.err || exit 1
  python -c "
import json; d = json.load(open('gpurun_out/bench_all_$w.json'))
print('$w', d['config']['workload'], '| Mrays/s %.0f Msamples/s %.0f ms/step %.2f kernel_ms %.2f frac %.3f rays/sample %.3f' % (d['value'], d['msamples_per_s'], d['ms_per_step'], d['kernel_ms_mean'], d['roofline']['frac'], d['rays_per_sample']))"
done
