#!/usr/bin/env python3
"""Where the wave time of k_trace goes: runs the diagnostic build (make -C raytracing_amd/csrc prof) and
prints the share of s_memtime ticks per section of the main loop.  RT_AMD_LIB must point at it."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["RT_AMD_LIB"] = os.path.join(ROOT, "tools", "librt_amd_prof.so")
sys.path.insert(0, ROOT)
import raytracing_amd as R

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 50
big = len(sys.argv) > 2 and sys.argv[2] == "c5"          # the 20 000-sphere scene at 1920x1080 instead of the final scene at 1200x800
W, H = (1920, 1080) if big else (1200, 800)
rt = R.Renderer(0)
rt.upload_scene(R.scene_grid(42, 75, 19996) if big else R.final_scene_table())
rt.set_camera(R.final_camera(W, H))
L = R.load_library()
buf = (C.c_ulonglong * 12)()
L.rt_debug_section_ticks(buf, 1)
rgb, irgb, rays, ms = rt.render(rt.params(W, H, spp, seed=0))
L.rt_debug_section_ticks(buf, 0)
names = ["hand-out", "ray setup+filter", "exact pass A", "exact pass B", "sparse sweep", "hit record", "scatter", "sky+store",
         "staging: camera rays", "staging: first hit over the tile list", "staging: first shading step", "staging: compaction, stores, wait"]
order = [0, 8, 9, 10, 11, 1, 2, 3, 4, 5, 6, 7]
tot = sum(buf[i] for i in order)
print(f"{rays} rays, kernel {ms:.2f} ms (timers add overhead)")
for i in order:
    print(f"{names[i]:>40s} {100.0 * buf[i] / tot:5.1f} %")
