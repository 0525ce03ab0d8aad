"""Scenes between 500 and 1536 spheres: records and filter operands in LDS (one 768-thread block per CU above 585 spheres) against
records from L2 (two blocks per CU).  RT_AMD_MAX_LDS_SPHERES picks the variant; kernel ms of a 1200x800x20 render."""
import os
import sys
import subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import torch
    import raytracing_amd as R
    rt = R.Renderer(0, camera_ray_lists=False)
    W, H = 1200, 800
    rt.set_camera(R.final_camera(W, H))
    out = torch.zeros((H, W, 3), dtype=torch.int32, device="cuda")
    for n in (487, 560, 600, 800, 1000, 1300, 1536):
        table = R.scene_grid(42, 20, n - 4)
        rt.upload_scene(table)
        best = 1e9
        for _ in range(4):
            rt.render_device(rt.params(W, H, 20, seed=0), 0, out.data_ptr())
            rays, ms = rt.stats()
            best = min(best, ms)
        print(f"n {len(table):5d}  {best:8.3f} ms  rays {rays}  checksum {int(out.to(torch.int64).sum())}", flush=True)
else:
    for limit in ("1536", "585", "1"):
        print(f"RT_AMD_MAX_LDS_SPHERES={limit}", flush=True)
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, RT_AMD_MAX_LDS_SPHERES=limit), check=True)
