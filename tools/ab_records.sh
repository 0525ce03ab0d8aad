#!/bin/bash
# Scalar-load records vs LDS-broadcast records (same kernel otherwise): bench + LDS counters.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for r in sgpr lds; do
  export RT_AMD_RECORDS=$r
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_rec_$r.json 2> gpurun_out/bench_rec_$r.err || exit 1
  python -c "
import json; d = json.load(open('gpurun_out/bench_rec_$r.json'))
print('$r Mrays/s %.0f ms/step %.2f frac %.3f checksum %s' % (d['value'], d['ms_per_step'], d['roofline']['frac'], d['image_checksum']))"
  timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_LDS --output-format csv -d gpurun_out/pmc_rec_$r -o p -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --spp 50 > gpurun_out/pmc_rec_$r.log 2>&1 || exit 1
done
