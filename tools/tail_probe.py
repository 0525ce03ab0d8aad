"""Kernel time of small renders of the final scene vs depth cap and sample count (tail / fixed-cost probe)."""
import sys
import numpy as np
import torch
import raytracing_amd as R

rt = R.Renderer(0)
rt.upload_scene(R.final_scene_table())
W, H = 1200, 800
rt.set_camera(R.final_camera(W, H))
out = torch.zeros((H, W, 3), dtype=torch.int32, device="cuda")
for depth in (50, 10, 3, 0):
    for spp in (1, 2, 5, 10, 20):
        p = rt.params(W, H, spp, seed=0, max_depth=depth)
        ts = []
        for rep in range(6):
            rt.render_device(p, 0, out.data_ptr())
            rays, ms = rt.stats()
            ts.append(ms)
        print("depth %2d spp %2d  kernel %.3f ms  rays %d  -> %.0f Mrays/s" % (depth, spp, min(ts[1:]), rays, rays / min(ts[1:]) / 1e3))
