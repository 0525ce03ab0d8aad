#!/bin/bash
# Config 5's scene (20 000 spheres, 1920x1080, 8 spp, one launch per pass) with the filter's operands as LDS tiles shared by a block
# (RT_AMD_LARGE_SCENES=tiles), the same three counter passes as tools/profile_round.sh makes for the default (streamed) form, and a
# kernel trace of both forms at 32 spp.  Output under gpurun_out/c5tiles/; tools/c5_tiles_summary.py writes profiles/<tag>_c5_tiles_pmc.txt.
out=gpurun_out/c5tiles
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
short="python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-beyond --workload c5 --spp 8"
pass() { name=$1; shift; timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $out/pmc_$name -o p -- $short > $out/pmc_$name.log 2>&1 || exit 1; }
for form in tiles stream; do
  export RT_AMD_LARGE_SCENES=$form
  pass ${form}_a SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU
  pass ${form}_b SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE
  pass ${form}_c TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_$form -o p -- python bench.py --no-cpu-baseline --no-beyond --workload c5 --spp 32 --steps 2 --warmup 1 > $out/trace_$form.log 2>&1 || exit 1
done
echo c5 tiles passes done
