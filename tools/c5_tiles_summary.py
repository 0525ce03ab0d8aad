#!/usr/bin/env python3
"""Summary of tools/c5_tiles_pmc.sh: the 20 000-sphere sweep with operand tiles in LDS against the streamed form, counters of k_trace side by side."""
import collections, csv, glob, os, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "c5tiles")
SIMDS = 1024
def counters(name):
    agg = collections.defaultdict(float)
    for path in glob.glob(os.path.join(src, f"pmc_{name}", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            if "k_trace" in r["Kernel_Name"]:
                agg[r["Counter_Name"]] += float(r["Counter_Value"])
    return agg
def trace_ms(form):
    for path in glob.glob(os.path.join(src, f"trace_{form}", "**", "*kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            if "k_trace" in r["Name"]:
                return float(r["AverageNs"]) / 1e6, int(r["Calls"])
    return None, 0
with open(os.path.join(root, "profiles", f"{tag}_c5_tiles_pmc.txt"), "w") as f:
    f.write("# tools/c5_tiles_pmc.sh + tools/c5_tiles_summary.py: k_trace on config 5's scene (20 000 spheres, 1920x1080; counter passes at 8 spp, one launch each; kernel trace at 32 spp)\n")
    f.write("# tiles = the filter's operands staged in LDS as tiles of 512 spheres per block, its twelve waves sweeping in step (RT_AMD_LARGE_SCENES=tiles); stream = every wave loads them from L2 for itself (default)\n")
    for form in ("tiles", "stream"):
        c = {}
        for p in "abc":
            c.update(counters(f"{form}_{p}"))
        ms, calls = trace_ms(form)
        f.write(f"== {form}: k_trace at 32 spp {ms:.1f} ms (mean of {calls} launches)\n")
        for k, v in sorted(c.items()):
            f.write(f"{k} {v:.0f}\n")
        wc = c.get("SQ_WAVE_CYCLES", 0.0); kc = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
        if wc and kc:
            f.write(f"# wave time split: active {c['SQ_ACTIVE_INST_ANY'] / wc:.2f} issue-stall {c['SQ_WAIT_INST_ANY'] / wc:.2f} waitcnt (incl. barrier) {c['SQ_WAIT_ANY'] / wc:.2f}\n")
            f.write(f"# kernel {kc:.4g} shader cycles; vector instructions x 2 cycles / SIMD cycles = {c['SQ_INSTS_VALU'] * 2 / SIMDS / kc:.3f}; matrix pipe busy {c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / SIMDS / kc:.3f}; "
                    f"LDS-issue stall share of wave time {c.get('SQ_WAIT_INST_LDS', 0) / wc:.4f}; vector-memory read instructions {c.get('SQ_INSTS_VMEM_RD', 0):.4g}, LDS instructions {c.get('SQ_INSTS_LDS', 0):.4g}\n")
        if c.get("TCC_HIT_sum", 0) + c.get("TCC_MISS_sum", 0) > 0:
            f.write(f"# L2: requests {c['TCC_HIT_sum'] + c['TCC_MISS_sum']:.4g}, hit rate {c['TCC_HIT_sum'] / (c['TCC_HIT_sum'] + c['TCC_MISS_sum']):.4f}; L1 requests {c.get('TCP_TOTAL_CACHE_ACCESSES_sum', 0):.4g}, of them to L2 {c.get('TCP_TCC_READ_REQ_sum', 0):.4g}\n")
print(open(os.path.join(root, "profiles", f"{tag}_c5_tiles_pmc.txt")).read())
