"""Kernel time of every rank's share of config 3 when the image is split over G GPUs -- by tiles (every G-th 8x8 tile of the work
order, the default partition) and in interleaved bands of `band` rows -- against 1/G of the single-GPU time: what the multi-GPU
bench can reach at best (the slowest rank sets the step time).  RT_AMD_PRIMARY=sweep times the brute-force path."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracing_amd as R

rt = R.Renderer(0)
rt.upload_scene(R.final_scene_table())
W, H, S = 1200, 800, 500
rt.set_camera(R.final_camera(W, H))
out = torch.zeros((H, W, 3), dtype=torch.int32, device="cuda")

def run(p, reps=3):
    best = 1e9
    for _ in range(reps):
        rt.render_device(p, 0, out.data_ptr())
        rays, ms = rt.stats()
        best = min(best, ms)
    return best, rays

full, rays_full = run(rt.params(W, H, S, seed=0))
print("1 GPU: %.2f ms, %d rays" % (full, rays_full))
G = int(sys.argv[1]) if len(sys.argv) > 1 else 8
def run_tiles(p, k, reps=3):
    n = rt.owned_tiles(p, k, G)
    t = torch.zeros((n, 64, 3), dtype=torch.int32, device="cuda")
    best = 1e9
    for _ in range(reps):
        rt.render_tiles_device(p, k, G, d_irgb_tiles=t.data_ptr())
        rays, ms = rt.stats()
        best = min(best, ms)
    return best, rays
res = [run_tiles(rt.params(W, H, S, seed=0), r) for r in range(G)]
ms = [x[0] for x in res]; rays = [x[1] for x in res]
print("G=%d tiles : ms per rank %s  (rays/1e6 %s)  slowest %.2f vs ideal %.2f -> %.2fx" %
      (G, " ".join("%.2f" % m for m in ms), " ".join("%.0f" % (r / 1e6) for r in rays), max(ms), full / G, full / max(ms)))
for band in (8, 4, 2, 1):
    res = [run(rt.params(W, H, S, seed=0, band_rows=band, band_first=r, band_step=G)) for r in range(G)]
    ms = [x[0] for x in res]; rays = [x[1] for x in res]
    print("G=%d band %d: ms per rank %s  (rays/1e6 %s)  slowest %.2f vs ideal %.2f -> %.2fx" %
          (G, band, " ".join("%.2f" % m for m in ms), " ".join("%.0f" % (r / 1e6) for r in rays), max(ms), full / G, full / max(ms)))
