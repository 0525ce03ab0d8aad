"""Wall time vs kernel time of band-by-band host renders (rt_render per band)."""
import time
import numpy as np
import raytracing_amd as R

rt = R.Renderer(0)
rt.upload_scene(R.final_scene_table())
W, H, S, B = 2560, 1440, 10, 64
rt.set_camera(R.final_camera(W, H))
bands = (H + B - 1) // B
for rep in range(2):
    t_all = time.perf_counter()
    rows = []
    for b in range(bands):
        p = rt.params(W, H, S, seed=0, band_rows=B, band_first=b, band_step=bands)
        t0 = time.perf_counter()
        rgb, irgb, rays, ms = rt.render(p)
        t1 = time.perf_counter()
        c = rt.counters()
        t2 = time.perf_counter()
        rows.append(((t1 - t0) * 1e3, ms, (t2 - t1) * 1e3))
    print("pass", rep, "total %.1f ms" % ((time.perf_counter() - t_all) * 1e3))
    for r in rows[:4] + rows[-2:]:
        print("  render wall %.2f ms, kernels %.2f ms, counters %.3f ms" % r)
p = rt.params(W, H, S, seed=0)
for rep in range(2):
    t0 = time.perf_counter(); rgb, irgb, rays, ms = rt.render(p); t1 = time.perf_counter()
    print("single call wall %.2f ms, kernels %.2f ms" % ((t1 - t0) * 1e3, ms))
