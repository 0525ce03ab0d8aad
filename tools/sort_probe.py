import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
import raytracing_amd as R
def morton(ix, iz):
    def spread(v):
        v = v.astype(np.uint64)
        v = (v | (v << 8)) & 0x00FF00FF
        v = (v | (v << 4)) & 0x0F0F0F0F
        v = (v | (v << 2)) & 0x33333333
        v = (v | (v << 1)) & 0x55555555
        return v
    return spread(ix) | (spread(iz) << 1)
rt = R.Renderer(0)
table = R.final_scene_table()
W, H, S = 1200, 800, 500
rt.set_camera(R.final_camera(W, H))
out = torch.zeros((H, W, 3), dtype=torch.int32, device="cuda")
def run(tab, label):
    rt.upload_scene(tab)
    best = 1e9
    for _ in range(4):
        rt.render_device(rt.params(W, H, S, seed=0), 0, out.data_ptr())
        rays, ms = rt.stats(); best = min(best, ms)
    print("%-28s %.2f ms  rays %d  checksum %d" % (label, best, rays, int(out.to(torch.int64).sum().item())), flush=True)
run(table, "upload order")
x, z = table[:, 0], table[:, 2]
small = np.abs(table[:, 3]) < 5
for cell in (1.0, 2.0, 4.0):
    ix = np.clip(((x + 16) / cell), 0, 65535).astype(np.int64); iz = np.clip(((z + 16) / cell), 0, 65535).astype(np.int64)
    key = morton(ix, iz).astype(np.int64)
    key[~small] = -1          # the ground first
    order = np.argsort(key, kind="stable")
    run(table[order], "morton xz, cell %.0f" % cell)
# rows of 2D blocks: block (bx, bz) of 6 x 5 cells
bx = np.floor((x + 11) / 6).astype(np.int64); bz = np.floor((z + 11) / 5).astype(np.int64)
key = bx * 100 + bz; key[~small] = -1
run(table[np.argsort(key, kind="stable")], "6 x 5 blocks")
rng = np.random.default_rng(0)
run(table[rng.permutation(len(table))], "random order")
