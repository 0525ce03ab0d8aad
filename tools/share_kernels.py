"""Kernel-by-kernel time of one rank's share of config 3 (rank 0 of 8), as a tile share and as a 1-row-band share, and of the whole
image through both entry points: run under `rocprofv3 --kernel-trace --stats`."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracing_amd as R

which = sys.argv[1] if len(sys.argv) > 1 else "tiles"
rt = R.Renderer(0)
rt.upload_scene(R.final_scene_table())
W, H, S, G = 1200, 800, 500, 8
rt.set_camera(R.final_camera(W, H))
out = torch.zeros((H, W, 3), dtype=torch.int32, device="cuda")
tiles = torch.zeros((rt.owned_tiles(rt.params(W, H, S), 0, 1), 64, 3), dtype=torch.int32, device="cuda")
for rep in range(6):
    if which == "tiles":
        rt.render_tiles_device(rt.params(W, H, S, seed=0), 0, G, d_irgb_tiles=tiles.data_ptr())
    elif which == "rows":
        rt.render_device(rt.params(W, H, S, seed=0, band_rows=1, band_first=0, band_step=G), 0, out.data_ptr())
    elif which == "whole_tiles":
        rt.render_tiles_device(rt.params(W, H, S, seed=0), 0, 1, d_irgb_tiles=tiles.data_ptr())
    else:
        rt.render_device(rt.params(W, H, S, seed=0), 0, out.data_ptr())
    rays, ms = rt.stats()
    print(which, rep, f"{ms:.3f} ms", rays, flush=True)
