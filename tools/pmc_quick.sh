#!/bin/bash
# Three counter passes of k_trace on the brute-force path (bench at 50 spp, one launch), printed as per-64-ray figures.
tag=${1:-quick}
out=gpurun_out/pmc_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
short="python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-beyond --spp 50"
pass() { name=$1; shift; timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $out/$name -o p -- $short > $out/$name.log 2>&1 || { tail -5 $out/$name.log; exit 1; }; }
pass a SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS
pass b SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_BF16
pass c SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE
python - $out <<'PY'
import csv, glob, sys, collections, os
out = sys.argv[1]
c = collections.defaultdict(float)
for path in glob.glob(os.path.join(out, "*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        if "k_trace" in r["Kernel_Name"]:
            c[r["Counter_Name"]] += float(r["Counter_Value"])
rays = 126863791.0
trips = rays / 64
cyc = c["GRBM_GUI_ACTIVE"] / 8
for k in sorted(c):
    print(f"{k:34s} {c[k]:.4g}   per 64 rays {c[k] / trips:9.1f}")
print(f"kernel cycles {cyc:.4g}; VALU insts x 2 cycles / SIMD cycles = {c['SQ_INSTS_VALU'] * 2 / 1024 / cyc:.3f}; MFMA busy / SIMD cycles = {c['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / cyc:.3f}")
print(f"active lanes per VALU inst {c['SQ_THREAD_CYCLES_VALU'] / c['SQ_INSTS_VALU']:.1f}; wave time: active {c['SQ_ACTIVE_INST_ANY'] / c['SQ_WAVE_CYCLES']:.2f} issue-stall {c['SQ_WAIT_INST_ANY'] / c['SQ_WAVE_CYCLES']:.2f} waitcnt {c['SQ_WAIT_ANY'] / c['SQ_WAVE_CYCLES']:.2f}")
PY
