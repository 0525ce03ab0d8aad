#!/bin/bash
# HBM traffic of the render kernels: FETCH_SIZE and WRITE_SIZE in separate passes (TCC slots), bench at 50 spp.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -o p -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --spp 50 > gpurun_out/pmc_fetch.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -o p -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --spp 50 > gpurun_out/pmc_write.log 2>&1 || exit 1
