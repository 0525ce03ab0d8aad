#!/bin/bash
# Two PMC passes over a short bench run (spp 50); writes CSVs under gpurun_out/pmc_<tag>{a,b}
tag=${1:-x}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS --output-format csv -d gpurun_out/pmc_${tag}a -o p -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --spp 50 > gpurun_out/pmc_${tag}a.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR --output-format csv -d gpurun_out/pmc_${tag}b -o p -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --spp 50 > gpurun_out/pmc_${tag}b.log 2>&1 || exit 1
