#!/usr/bin/env python3
"""Camera rays through the per-tile sphere lists against camera rays swept like every other ray (RT_AMD_PRIMARY=sweep, read when a
context is created): kernel ms of a 1200x800 image at 20 samples for the first n spheres of the final scene, n = 4 ... 487."""
import os
import sys
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracing_amd as R

table = R.final_scene_table()
W, H = 1200, 800
out = torch.zeros((H, W, 3), dtype=torch.int32, device="cuda")
def renderer(primary):
    if primary == "sweep":
        os.environ["RT_AMD_PRIMARY"] = "sweep"
    else:
        os.environ.pop("RT_AMD_PRIMARY", None)
    rt = R.Renderer(0)
    rt.set_camera(R.final_camera(W, H))
    return rt
rts = {"lists": renderer("lists"), "sweep": renderer("sweep")}
for n in (4, 8, 16, 32, 64, 128, 256, len(table)):
    res = {}
    for name, rt in rts.items():
        rt.upload_scene(table[:n])
        p = rt.params(W, H, 20, seed=0)
        best = 1e9
        for _ in range(4):
            rt.render_device(p, 0, out.data_ptr())
            rays, ms = rt.stats()
            best = min(best, ms)
        res[name] = (best, rays, int(out.to(torch.int64).sum().item()))
    same = res["lists"][1:] == res["sweep"][1:]
    print(f"n {n:4d}  lists {res['lists'][0]:7.3f} ms  sweep {res['sweep'][0]:7.3f} ms  ratio {res['lists'][0] / res['sweep'][0]:.3f}  identical {same}")
