"""One wave alone on the chip, 8 ... 64 of its lanes on 65-query paths (a camera inside a mirror sphere): what a trip of the main
loop costs such a wave and where it goes (section-timer build, s_memtime ticks of 10 ns).  This is the trip that the last waves of
a launch make over and over while the rest of the chip is idle."""
import ctypes as C
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["RT_AMD_LIB"] = os.path.join(ROOT, "tools", os.environ.get("RT_PROF_LIB", "librt_amd_prof.so"))
sys.path.insert(0, ROOT)
import raytracing_amd as R

rt = R.Renderer(0)
table = np.vstack([R.final_scene_table(), [[100.0, 50.0, 100.0, -1.0, R.METAL, 0.9, 0.9, 0.9, 0.0]]])
rt.upload_scene(table)
rt.set_camera(R.Camera((100.0, 50.0, 100.0), (101.0, 50.2, 100.3), (0.0, 1.0, 0.0), 40.0, 1.0, 0.0, 1.0))
L = R.load_library()
buf = (C.c_ulonglong * 12)()
names = ["hand-out", "ray setup+filter", "exact pass A", "exact pass B", "sparse sweep", "hit record", "scatter", "sky+store", "staging: camera", "staging: first hit", "staging: shading", "staging: rest"]
for h in (1, 2, 4, 8):
    rt.render(rt.params(8, h, 1, seed=0, max_depth=64))
    L.rt_debug_section_ticks(buf, 1)
    rgb, irgb, rays, ms = rt.render(rt.params(8, h, 1, seed=0, max_depth=64))
    L.rt_debug_section_ticks(buf, 0)
    lanes = 8 * h
    trips = rays / lanes - 1          # the first query of every path is answered in staging
    line = "  ".join(f"{names[i]} {buf[i] / trips / 100.0:.2f}" for i in range(8) if buf[i] / trips >= 5)
    print(f"{lanes:2d} live lanes: kernel {ms * 1e3:.0f} us, {trips:.0f} trips, {ms * 1e3 / trips:.2f} us per trip | us per trip: {line}")
