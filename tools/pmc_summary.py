#!/usr/bin/env python3
"""Turns the rocprofv3 output of tools/profile_round.sh (gpurun_out/<tag>/) into the files kept under profiles/:
<tag>_kernel_stats.csv (kernel-trace statistics of the default bench command), <tag>_pmc_k_trace.txt (raw counters of k_trace
with the derived figures) and <tag>_pmc_k_trace.json (per-ray figures bench.py reads for roofline.traffic and executed_frac).
The camera rays of a launch come from k_primary, which runs before k_trace: executed operations and HBM bytes per ray include it
(its counters are listed apart), the cycle-based figures (issue share, wave time split, matrix pipe) are k_trace's.

Units as observed on gfx950: SQ_*_CYCLES / SQ_ACTIVE_INST_* / SQ_WAIT_* count in units of four cycles summed over waves;
SQ_INSTS_* count wave-instructions; FETCH_SIZE under-reports wide coalesced reads by a factor two (calibrated on k_resolve, whose
bytes are known); WRITE_SIZE is exact; both in KB.  GRBM_GUI_ACTIVE is summed over the eight XCDs.
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", tag)
dst = os.path.join(root, "profiles")
SIMDS, LANES = 1024, 64


def counters(name, kernel="k_trace"):
    agg = collections.defaultdict(float)
    launches = collections.Counter()
    for path in glob.glob(os.path.join(src, f"pmc_{name}", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            if kernel in r["Kernel_Name"]:
                agg[r["Counter_Name"]] += float(r["Counter_Value"])
                launches[r["Counter_Name"]] += 1
    return agg, launches


bench = json.load(open(os.path.join(src, "bench.json")))
rays_per_sample = bench["rays_per_sample"]
samples = 1200 * 800 * 50                      # the counter passes run the bench at 50 samples per pixel, one launch
rays = samples * rays_per_sample
c, cp = {}, {}
for name in "abc":
    a, _ = counters(name)
    c.update(a)
    a, _ = counters(name, "k_primary")
    cp.update(a)
fetch, _ = counters("fetch")
write, _ = counters("write")
pfetch, _ = counters("fetch", "k_primary")
pwrite, _ = counters("write", "k_primary")
rfetch, _ = counters("fetch", "k_resolve")
shutil.copy(os.path.join(src, "bench.json"), os.path.join(dst, f"{tag}_bench_line.json"))

kernel_cycles = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0                 # shader cycles the launch lasted
insts = c["SQ_INSTS_VALU"]
lanes = c["SQ_THREAD_CYCLES_VALU"] / insts                           # mean active lanes per vector instruction
fma, add, mul, trans = (c.get(k, 0.0) for k in ("SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_TRANS_F32"))
# executed FP32 operations per ray, two ways: every FP32 instruction at all 64 lanes (upper bound: what the ALU was occupied
# for), and scaled by the mean active-lane share of all vector instructions (the filter's FMAs run with every lane of a live
# ray active, so the truth lies between)
flop_wave = (2 * fma + add + mul + trans) * LANES
pflop_wave = (2 * cp.get("SQ_INSTS_VALU_FMA_F32", 0.0) + cp.get("SQ_INSTS_VALU_ADD_F32", 0.0) + cp.get("SQ_INSTS_VALU_MUL_F32", 0.0) + cp.get("SQ_INSTS_VALU_TRANS_F32", 0.0)) * LANES
executed_flop_per_ray_upper = (flop_wave + pflop_wave) / rays
executed_flop_per_ray_lanes = flop_wave * (lanes / LANES) / rays + pflop_wave / rays
hbm_trace = (2.0 * fetch.get("FETCH_SIZE", 0.0) + write.get("WRITE_SIZE", 0.0)) * 1024.0
hbm_primary = (2.0 * pfetch.get("FETCH_SIZE", 0.0) + pwrite.get("WRITE_SIZE", 0.0)) * 1024.0
hbm_bytes = hbm_trace + hbm_primary
summary = {
    "tag": tag, "workload": "1200x800 final scene, 50 spp (one launch of k_trace per counter pass)", "spheres": 487,
    "rays": rays, "hbm_bytes_per_ray": hbm_bytes / rays, "hbm_bytes_per_ray_k_trace": hbm_trace / rays, "hbm_bytes_per_ray_k_primary": hbm_primary / rays,
    "k_primary_valu_insts_per_64_rays": cp.get("SQ_INSTS_VALU", 0.0) / (rays / 64),
    "valu_insts_per_64_rays": insts / (rays / 64), "mean_active_lanes": lanes,
    "fma_f32_per_64_rays": fma / (rays / 64), "add_f32_per_64_rays": add / (rays / 64), "mul_f32_per_64_rays": mul / (rays / 64),
    "trans_f32_per_64_rays": trans / (rays / 64), "int32_per_64_rays": c.get("SQ_INSTS_VALU_INT32", 0.0) / (rays / 64),
    "executed_flop_per_ray": executed_flop_per_ray_upper, "executed_flop_per_ray_active_lanes": executed_flop_per_ray_lanes,
    "valu_utilization": lanes / LANES,
    "kernel_cycles": kernel_cycles,
}
if kernel_cycles > 0:
    # VALUBusy as rocprof's gfx9 formula defines it (4 cycles per vector instruction): on this part a plain FP32 instruction holds
    # the pipe for 2, so the figure can exceed 1; valu_issue_frac prices every instruction at 2 cycles instead (a lower bound of
    # the share of SIMD cycles with a vector instruction in the pipe)
    summary["valu_busy_gfx9_formula"] = c["SQ_ACTIVE_INST_VALU"] * 4 / SIMDS / kernel_cycles
    summary["valu_issue_frac"] = insts * 2 / SIMDS / kernel_cycles
# kernel-trace statistics, one command per launch group (tools/profile_round.sh): trace_c3 = the headline workload alone (config 3, brute
# force), trace_c1 = the short launch, trace_lists = config 3 with the camera-ray shortcut, trace_c5 = the 20 000-sphere scene at 32 spp,
# trace_default = the default command (headline + config-1 line + beyond_brute_force launches in one process)
groups = {}
for name in ("default", "c3", "c1", "lists", "c5"):
    st = glob.glob(os.path.join(src, f"trace_{name}", "**", "*kernel_stats.csv"), recursive=True)
    if not st:
        continue
    shutil.copy(st[0], os.path.join(dst, f"{tag}_kernel_stats_{name}.csv"))
    for r in csv.DictReader(open(st[0])):
        if "k_trace" in r["Name"]:
            groups.setdefault(name, {}).update({"launches": int(r["Calls"]), "mean_ms": float(r["AverageNs"]) * 1e-6, "min_ms": float(r["MinNs"]) * 1e-6})
        if r["Name"].startswith("k_primary"):
            groups.setdefault(name, {}).update({"k_primary_mean_ms": float(r["AverageNs"]) * 1e-6})
        if r["Name"].startswith("k_resolve"):
            groups.setdefault(name, {}).update({"k_resolve_mean_ms": float(r["AverageNs"]) * 1e-6})
summary["kernel_trace_groups"] = groups
if "c3" in groups:
    summary["kernel_trace_headline_mean_ms"] = groups["c3"]["mean_ms"]
    summary["kernel_trace_headline_step_ms"] = groups["c3"]["mean_ms"] + groups["c3"].get("k_primary_mean_ms", 0.0) + groups["c3"].get("k_resolve_mean_ms", 0.0)
summary["bench_kernel_ms_mean"] = bench.get("kernel_ms_mean")
with open(os.path.join(dst, f"{tag}_pmc_k_trace.txt"), "w") as f:
    f.write(f"# k_trace, bench.py --spp 50 --steps 1 (one launch per pass, {rays:.4g} rays); passes a, b, c of tools/profile_round.sh\n")
    for k, v in sorted(c.items()):
        f.write(f"{k} {v:.0f}\n")
    f.write(f"FETCH_SIZE_KB {fetch.get('FETCH_SIZE', 0):.0f}\nWRITE_SIZE_KB {write.get('WRITE_SIZE', 0):.0f}\nk_resolve_FETCH_SIZE_KB {rfetch.get('FETCH_SIZE', 0):.0f} (known bytes: {samples * 12 / 1024:.0f} KB)\n")
    wc = c["SQ_WAVE_CYCLES"]
    f.write(f"# per 64 rays: VALU {summary['valu_insts_per_64_rays']:.0f} wave-instructions (FMA_F32 {summary['fma_f32_per_64_rays']:.0f}, ADD_F32 {summary['add_f32_per_64_rays']:.0f}, "
            f"MUL_F32 {summary['mul_f32_per_64_rays']:.0f}, TRANS_F32 {summary['trans_f32_per_64_rays']:.0f}, INT32 {summary['int32_per_64_rays']:.0f}), "
            f"SALU {c['SQ_INSTS_SALU'] / (rays / 64):.0f}, SMEM {c['SQ_INSTS_SMEM'] / (rays / 64):.0f}, LDS {c['SQ_INSTS_LDS'] / (rays / 64):.0f}\n")
    f.write(f"# VALUUtilization (active lanes per vector instruction): {lanes:.1f} / 64 = {lanes / 64:.3f}\n")
    if kernel_cycles > 0:
        f.write(f"# kernel duration {kernel_cycles:.4g} shader cycles (GRBM_GUI_ACTIVE / 8 XCDs); VALUBusy by the gfx9 formula (4 cycles per instruction) "
                f"{summary['valu_busy_gfx9_formula']:.3f}; at 2 cycles per instruction {summary['valu_issue_frac']:.3f} of SIMD cycles\n")
    f.write(f"# wave time split: active {c['SQ_ACTIVE_INST_ANY'] / wc:.2f} issue-stall {c['SQ_WAIT_INST_ANY'] / wc:.2f} waitcnt {c['SQ_WAIT_ANY'] / wc:.2f}; waves {c['SQ_WAVES']:.0f}\n")
    f.write(f"# executed FP32 operations per ray: {executed_flop_per_ray_upper:.0f} with all 64 lanes counted, {executed_flop_per_ray_lanes:.0f} scaled by the active-lane share; "
            f"algorithmic (reference) work per ray: {18 * 487 + 150}\n")
    f.write(f"# HBM bytes per ray: {summary['hbm_bytes_per_ray']:.1f} (FETCH_SIZE x 2 + WRITE_SIZE; k_trace {hbm_trace / rays:.1f}, k_primary {hbm_primary / rays:.1f})\n")
    if cp:
        f.write("# k_primary (camera rays of the launch, before k_trace): " + ", ".join(f"{k} {v:.0f}" for k, v in sorted(cp.items()) if k.startswith("SQ_INSTS") or k in ("SQ_WAVES", "GRBM_GUI_ACTIVE"))
                + f"; FETCH_SIZE_KB {pfetch.get('FETCH_SIZE', 0):.0f} WRITE_SIZE_KB {pwrite.get('WRITE_SIZE', 0):.0f}\n")
    for name, g in groups.items():
        f.write(f"# rocprofv3 --kernel-trace, command group {name}: {g['launches']} k_trace launches, mean {g['mean_ms']:.3f} ms, min {g['min_ms']:.3f} ms"
                + (f"; k_primary mean {g['k_primary_mean_ms']:.3f} ms" if "k_primary_mean_ms" in g else "") + (f"; k_resolve mean {g['k_resolve_mean_ms']:.3f} ms" if "k_resolve_mean_ms" in g else "")
                + f" ({tag}_kernel_stats_{name}.csv)\n")
    f.write(f"# bench.py's HIP-event figure for the headline (spans k_primary, k_trace and k_resolve): {bench.get('kernel_ms_mean'):.2f} ms\n")
    dpass, _ = counters("d")
    if dpass:
        trips = rays / 64
        f.write("# matrix pipe: " + ", ".join(f"{k} {v:.4g} ({v / trips:.1f} per 64 rays)" for k, v in sorted(dpass.items())) + "\n")
        if "SQ_INSTS_VALU_MFMA_MOPS_BF16" in dpass:
            # one count = 512 floating-point operations (a 32x32x8 instruction, 16 384 operations, counts 32)
            summary["mfma_bf16_flop_per_ray"] = dpass["SQ_INSTS_VALU_MFMA_MOPS_BF16"] * 512.0 / rays
            f.write(f"# bf16 operations on the matrix pipe per ray: {summary['mfma_bf16_flop_per_ray']:.0f} (SQ_INSTS_VALU_MFMA_MOPS_BF16 x 512)\n")
        if kernel_cycles > 0 and "SQ_VALU_MFMA_BUSY_CYCLES" in dpass:
            summary["mfma_busy_frac"] = dpass["SQ_VALU_MFMA_BUSY_CYCLES"] / SIMDS / kernel_cycles
            f.write(f"# SQ_VALU_MFMA_BUSY_CYCLES / (SIMDs x kernel cycles) = {summary['mfma_busy_frac']:.3f}\n")
# config 5: what the waves of the 20 000-sphere sweep wait for (filter operands and exact records from L2, nothing of the scene in LDS)
c5 = {}
for name in ("c5a", "c5b", "c5c"):
    a5, _ = counters(name)
    c5.update(a5)
if c5:
    with open(os.path.join(dst, f"{tag}_c5_pmc.txt"), "w") as f:
        f.write("# k_trace on config 5's scene (20 000 spheres, 1920x1080, 8 spp, one launch per pass): passes c5a, c5b, c5c of tools/profile_round.sh\n")
        for k, v in sorted(c5.items()):
            f.write(f"{k} {v:.0f}\n")
        wc5 = c5.get("SQ_WAVE_CYCLES", 0.0)
        if wc5:
            f.write(f"# wave time split: active {c5['SQ_ACTIVE_INST_ANY'] / wc5:.2f} issue-stall {c5['SQ_WAIT_INST_ANY'] / wc5:.2f} waitcnt {c5['SQ_WAIT_ANY'] / wc5:.2f}\n")
        kc5 = c5.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
        if kc5:
            f.write(f"# kernel {kc5:.4g} shader cycles; vector instructions x 2 cycles / SIMD cycles = {c5['SQ_INSTS_VALU'] * 2 / SIMDS / kc5:.3f}; "
                    f"matrix pipe busy {c5.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / SIMDS / kc5:.3f}; LDS-issue stall share of wave time {c5.get('SQ_WAIT_INST_LDS', 0) / wc5:.4f}\n")
        if c5.get("TCC_HIT_sum", 0) + c5.get("TCC_MISS_sum", 0) > 0:
            f.write(f"# L2: hit rate {c5['TCC_HIT_sum'] / (c5['TCC_HIT_sum'] + c5['TCC_MISS_sum']):.4f}; L1 requests {c5.get('TCP_TOTAL_CACHE_ACCESSES_sum', 0):.4g}, of them to L2 {c5.get('TCP_TCC_READ_REQ_sum', 0):.4g}\n")
with open(os.path.join(dst, f"{tag}_pmc_k_trace.json"), "w") as f:
    json.dump(summary, f, indent=1)
print(json.dumps(summary, indent=1))
