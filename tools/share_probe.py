"""Same number of samples, different shapes: whole image at S/8 samples per pixel vs one of eight interleaved shares at S."""
import torch
import raytracing_amd as R

rt = R.Renderer(0)
rt.upload_scene(R.final_scene_table())
W, H = 1200, 800
rt.set_camera(R.final_camera(W, H))
out = torch.zeros((H, W, 3), dtype=torch.int32, device="cuda")

def run(p, reps=4):
    best = 1e9
    for _ in range(reps):
        rt.render_device(p, 0, out.data_ptr())
        rays, ms = rt.stats()
        best = min(best, ms)
    return best, rays

for name, p in [
    ("whole image, 62 spp", rt.params(W, H, 62, seed=0)),
    ("whole image, 63 spp", rt.params(W, H, 63, seed=0)),
    ("share 0 of 8 (4-row bands), 500 spp", rt.params(W, H, 500, seed=0, band_rows=4, band_first=0, band_step=8)),
    ("share 0 of 8 (8-row bands, 100 bands -> 13 bands), 500 spp", rt.params(W, H, 500, seed=0, band_rows=8, band_first=0, band_step=8)),
    ("share 0 of 8 (100-row block), 500 spp", rt.params(W, H, 500, seed=0, band_rows=100, band_first=0, band_step=8)),
    ("share 3 of 8 (100-row block), 500 spp", rt.params(W, H, 500, seed=0, band_rows=100, band_first=3, band_step=8)),
    ("share 0 of 8 (4-row bands), 1000 spp", rt.params(W, H, 1000, seed=0, band_rows=4, band_first=0, band_step=8)),
    ("whole image, 500 spp", rt.params(W, H, 500, seed=0)),
    ("whole image, 1000 spp", rt.params(W, H, 1000, seed=0)),
]:
    ms, rays = run(p)
    print("%-62s %8.2f ms  %10d rays  %7.0f Mrays/s" % (name, ms, rays, rays / ms / 1e3))
