#!/usr/bin/env python3
"""Pixel-exact comparison of the literal kernel (RT_MODE_PIXEL) with the two filter variants of the megakernel."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracing_amd as R
w, h, spp = 1200, 800, int(sys.argv[1]) if len(sys.argv) > 1 else 100
def render(filt, mode):
    os.environ["RT_AMD_FILTER"] = filt
    rt = R.Renderer(0)
    rt.upload_scene(R.final_scene_table()); rt.set_camera(R.final_camera(w, h))
    rgb, irgb, rays, ms = rt.render(rt.params(w, h, spp, seed=0, mode=mode))
    rt.close()
    return rgb, rays, ms
ref, rays0, ms0 = render("disc", R.MODE_PIXEL)
print("literal kernel rays", rays0, "ms", ms0)
for f in ("disc", "plane"):
    img, rays, ms = render(f, R.MODE_MEGAKERNEL)
    bad = np.argwhere((img.view(np.uint32) != ref.view(np.uint32)).any(axis=2))
    print(f, "rays", rays, "ms %.1f" % ms, "mismatching pixels", len(bad), bad[:8].tolist())
