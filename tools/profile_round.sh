#!/bin/bash
# The round's profile of the default bench command, run on the GPU box from the repository root: the bench line, rocprofv3 kernel-trace
# statistics of the same command, and the counter passes of k_trace (one rocprofv3 run per pass, bench at 50 spp, one launch).
# Everything lands under gpurun_out/<tag>/; tools/pmc_summary.py <tag> turns it into profiles/<tag>_*.
tag=${1:-r02}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 python bench.py > $out/bench.json 2> $out/bench.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o p -- python bench.py --no-cpu-baseline > $out/trace.log 2>&1 || exit 1
short="python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-sweep-only --spp 50"
pass() { name=$1; shift; timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $out/pmc_$name -o p -- $short > $out/pmc_$name.log 2>&1 || exit 1; }
pass a SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS
pass b SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR
pass c SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE
pass fetch FETCH_SIZE
pass write WRITE_SIZE
echo profile passes done
