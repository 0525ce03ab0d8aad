"""Kernel time of the whole 1200x800 final-scene image against samples per pixel: is the cost a line (fixed + slope)?"""
import torch
import raytracing_amd as R
rt = R.Renderer(0)
rt.upload_scene(R.final_scene_table())
W, H = 1200, 800
rt.set_camera(R.final_camera(W, H))
out = torch.zeros((H, W, 3), dtype=torch.int32, device="cuda")
prev = None
import os, sys
for spp in (tuple(int(v) for v in sys.argv[1:]) or (1, 2, 4, 8, 16, 31, 62, 125, 250, 500, 1000)):
    p = rt.params(W, H, spp, seed=0)
    best = 1e9
    for _ in range(4):
        rt.render_device(p, 0, out.data_ptr())
        rays, ms = rt.stats()
        best = min(best, ms)
    inc = "" if prev is None else "  marginal %.4f ms per spp" % ((best - prev[1]) / (spp - prev[0]))
    print("spp %4d  %8.3f ms  %6.0f Mrays/s%s" % (spp, best, rays / best / 1e3, inc))
    prev = (spp, best)
