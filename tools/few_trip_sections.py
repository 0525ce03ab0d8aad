"""tools/few_trip_probe.py's one-path case on the section-timer build: s_memtime ticks (100 MHz on this part) per section of the
main loop, per trip, for a wave that holds a single 65-query path."""
import ctypes as C
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["RT_AMD_LIB"] = os.path.join(ROOT, "tools", "librt_amd_prof.so")
sys.path.insert(0, ROOT)
import raytracing_amd as R

rt = R.Renderer(0)
table = np.vstack([R.final_scene_table(), [[100.0, 50.0, 100.0, -1.0, R.METAL, 0.9, 0.9, 0.9, 0.0]]])
rt.upload_scene(table)
rt.set_camera(R.Camera((100.0, 50.0, 100.0), (101.0, 50.2, 100.3), (0.0, 1.0, 0.0), 40.0, 1.0, 0.0, 1.0))
L = R.load_library()
buf = (C.c_ulonglong * 12)()
names = ["hand-out", "ray setup+filter", "exact pass A", "exact pass B", "sparse sweep (incl. loop top)", "hit record", "scatter", "sky+store"]
for w in (1, 4):
    rt.render(rt.params(w, 1, 1, seed=0, max_depth=64))
    L.rt_debug_section_ticks(buf, 1)
    rgb, irgb, rays, ms = rt.render(rt.params(w, 1, 1, seed=0, max_depth=64))
    L.rt_debug_section_ticks(buf, 0)
    trips = rays / w
    print(f"{w} live lanes: kernel {ms * 1e3:.1f} us, {trips:.0f} trips")
    for i in range(8):
        print(f"   {names[i]:>30s} {buf[i] / trips:8.1f} ticks per trip")
