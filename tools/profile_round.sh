#!/bin/bash
# The round's profile, run on the GPU box from the repository root: the default bench line; rocprofv3 kernel-trace statistics of the
# default command and of ONE launch group per command (headline = config 3 brute force, config 1, camera-ray lists, config 5); and the
# counter passes of k_trace (one rocprofv3 run per pass, bench at 50 spp, one launch).  Everything lands under gpurun_out/<tag>/;
# tools/pmc_summary.py <tag> turns it into profiles/<tag>_*.
tag=${1:-r03}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 python bench.py > $out/bench.json 2> $out/bench.err || exit 1
trace() { name=$1; shift; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_$name -o p -- python bench.py --no-cpu-baseline "$@" > $out/trace_$name.log 2>&1 || exit 1; }
trace default
trace c3 --no-beyond --no-config1
trace c1 --no-beyond --workload c1 --steps 20 --warmup 3
trace lists --no-beyond --no-config1 --camera-ray-lists
trace c5 --no-beyond --workload c5 --spp 32 --steps 2 --warmup 1
short="python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-beyond --spp 50"
pass() { name=$1; shift; timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $out/pmc_$name -o p -- $short > $out/pmc_$name.log 2>&1 || exit 1; }
pass a SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS
pass b SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR
pass c SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE
pass d SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD
pass fetch FETCH_SIZE
pass write WRITE_SIZE
# config 5 (20 000 spheres: filter operands and exact records come from L2, nothing of the scene is staged in LDS): what its waves wait for
short="python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-beyond --workload c5 --spp 8"
pass c5a SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU
pass c5b SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE
pass c5c TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum
# the bench line that gets committed is taken AFTER the counters are summarised: bench.py reads the summary for its traffic and executed-work fields
python tools/pmc_summary.py $tag > $out/summary.log 2>&1 || exit 1
timeout -k 10 300 python bench.py > $out/bench.json 2> $out/bench.err || exit 1
echo profile passes done
