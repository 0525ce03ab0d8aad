import csv, collections, sys
tag = sys.argv[1]
agg = collections.defaultdict(float)
for part in "ab":
    for r in csv.DictReader(open(f"gpurun_out/pmc_{tag}{part}/p_counter_collection.csv")):
        if "k_trace" in r["Kernel_Name"] or "k_render" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in sorted(agg.items()):
    print(f"{k} {v:.0f}")
rays = 960000 * 50 * 2.6427
it = rays / 64
print(f"# VALU wave-instr per 64 rays: {agg['SQ_INSTS_VALU'] / it:.0f}; mean active lanes {agg['SQ_THREAD_CYCLES_VALU'] / agg['SQ_INSTS_VALU']:.1f}/64; "
      f"SALU per 64 rays {agg['SQ_INSTS_SALU'] / it:.0f}; SMEM {agg['SQ_INSTS_SMEM'] / it:.0f}; LDS {agg['SQ_INSTS_LDS'] / it:.0f}")
print(f"# wave time split: active {agg['SQ_ACTIVE_INST_ANY'] / agg['SQ_WAVE_CYCLES']:.2f} issue-stall {agg['SQ_WAIT_INST_ANY'] / agg['SQ_WAVE_CYCLES']:.2f} waitcnt {agg['SQ_WAIT_ANY'] / agg['SQ_WAVE_CYCLES']:.2f}; waves {agg['SQ_WAVES']:.0f}")
