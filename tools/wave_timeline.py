#!/usr/bin/env python3
"""When do the persistent waves of k_trace stop?  Diagnostic build (make -C raytracing_amd/csrc prof, the -DRT_WAVE_TIMELINE library): per wave the
s_memrealtime stamps at entry, at the moment it first found the work counter dry, and at exit (100 MHz ticks)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["RT_AMD_LIB"] = os.path.join(ROOT, "tools", "librt_amd_timeline.so")
sys.path.insert(0, ROOT)
import raytracing_amd as R

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 10
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 50
ranks = int(sys.argv[3]) if len(sys.argv) > 3 else 1          # render only rank 0's share of an image split over this many GPUs
rt = R.Renderer(0)
rt.upload_scene(R.final_scene_table())
rt.set_camera(R.final_camera(1200, 800))
L = R.load_library()
buf = np.zeros((10, 8192), dtype=np.uint64)
for rep in range(2):
    band = next(b for b in (8, 4, 2, 1) if (800 // b) % ranks == 0)
    if os.environ.get("RT_SHARE_TILES") == "1":          # rank 0's TILE share instead of its row bands
        import torch
        pt = rt.params(1200, 800, spp, seed=0, max_depth=depth)
        tiles = torch.zeros((rt.owned_tiles(pt, 0, ranks), 64, 3), dtype=torch.int32, device="cuda")
        rt.render_tiles_device(pt, 0, ranks, d_irgb_tiles=tiles.data_ptr())
        rays, ms = rt.stats()
    else:
        rgb, irgb, rays, ms = rt.render(rt.params(1200, 800, spp, seed=0, max_depth=depth, band_rows=band, band_first=0, band_step=ranks))
L.rt_debug_wave_times(buf.ctypes.data_as(C.POINTER(C.c_ulonglong)))
live = buf[2] > 0
t0 = buf[0][live].min()
enter, dry, exit_, trips = [(buf[k][live].astype(np.int64) - int(t0)) / 100.0 for k in range(3)] + [buf[3][live]]   # microseconds
print(f"spp {spp} depth {depth}: {rays} rays, kernel {ms:.3f} ms, {live.sum()} waves")
print("entry   us: min %.0f  median %.0f  max %.0f" % (enter.min(), np.median(enter), enter.max()))
print("dry     us: min %.0f  median %.0f  max %.0f" % (dry.min(), np.median(dry), dry.max()))
print("exit    us: min %.0f  p10 %.0f  median %.0f  p90 %.0f  p99 %.0f  max %.0f" % (exit_.min(), *np.percentile(exit_, [10, 50, 90, 99]), exit_.max()))
d = exit_ - dry
print("drain   us: median %.0f  p90 %.0f  p99 %.0f  max %.0f;  trips after dry: median %.0f p90 %.0f max %.0f" %
      (np.median(d), *np.percentile(d, [90, 99]), d.max(), np.median(trips), np.percentile(trips, 90), trips.max()))
edges = np.linspace(0, exit_.max(), 21)
alive = [(exit_ > e).sum() for e in edges]
print("waves still running at t (us):", ", ".join(f"{e:.0f}:{a}" for e, a in zip(edges, alive)))
st, sk = buf[4][live].astype(np.float64), buf[5][live].astype(np.float64) / 100.0
print("sparse trips: total %d, mean %.2f us per trip; waves with > 20 sparse trips: %d, their mean trip %.2f us" %
      (st.sum(), sk.sum() / max(st.sum(), 1), (st > 20).sum(), sk[st > 20].sum() / max(st[st > 20].sum(), 1)))
fs = (buf[6][live].astype(np.int64) - int(t0)) / 100.0
fd = (buf[7][live].astype(np.int64) - int(t0)) / 100.0
print("first sweep starts us: min %.0f median %.0f max %.0f;  second trip's sweep starts us: median %.0f max %.0f" % (fs.min(), np.median(fs), fs.max(), np.median(fd), fd.max()))
# the end shared by the waves of a block (k_trace with RT_TAIL_PATHS > 0): when it begins, how long it lasts, how many bounces it makes
sh = buf[8][live].astype(np.int64)
if (sh > 0).any():
    began = (sh[sh > 0] - int(t0)) / 100.0
    lasted = exit_[sh > 0] - began
    nb = buf[9][live][sh > 0].astype(np.float64)
    print("shared end: begins us: min %.0f median %.0f p90 %.0f max %.0f;  lasts us: median %.0f p90 %.0f max %.0f;  bounces: median %.0f p90 %.0f max %.0f;  us per bounce: median %.1f" %
          (began.min(), np.median(began), np.percentile(began, 90), began.max(), np.median(lasted), np.percentile(lasted, 90), lasted.max(),
           np.median(nb), np.percentile(nb, 90), nb.max(), np.median(lasted[nb > 0] / nb[nb > 0])))
    print("            from the counter running dry to the shared end us: median %.0f p90 %.0f max %.0f" % tuple(np.percentile(began - dry[sh > 0], [50, 90, 100])))
# the stragglers: for the waves that exit last, how the time after the counter ran dry splits into trips with the ordinary sweep and trips with the sparse one
order = np.argsort(-exit_)[:12]
print("last waves: exit us | dry us | trips after dry | of them sparse | us in sparse trips | us in ordinary trips | us per ordinary trip")
for w in order:
    full = trips[w] - st[w]
    t_full = (exit_[w] - dry[w]) - sk[w]
    print("   %7.0f | %6.0f | %3d | %3d | %6.0f | %6.0f | %5.1f" % (exit_[w], dry[w], trips[w], st[w], sk[w], t_full, t_full / max(full, 1)))
late = exit_ > np.percentile(exit_, 99)
print("slowest 1 %% of waves: mean trips after dry %.1f, of them sparse %.1f; mean us in sparse %.0f, in ordinary %.0f" %
      (trips[late].mean(), st[late].mean(), sk[late].mean(), ((exit_ - dry)[late] - sk[late]).mean()))
