"""Latency of one bounce of a wave that holds only a few paths (the end of every launch): a camera inside a mirror shell added to
the 487-sphere final scene, so that every path runs to the depth cap; images of 1..64 pixels, one sample each, are one wave with that
many live lanes.  Kernel time / (depth cap + 1) = time per trip of the main loop (sweep + shade)."""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracing_amd as R

rt = R.Renderer(0)
# the final scene plus a mirror shell (negative radius: the normal points inwards, so a fuzz-free metal reflects for ever)
# around the camera, far from everything else: every path bounces until the depth cap
table = np.vstack([R.final_scene_table(), [[100.0, 50.0, 100.0, -1.0, R.METAL, 0.9, 0.9, 0.9, 0.0]]])
rt.upload_scene(table)
rt.set_camera(R.Camera((100.0, 50.0, 100.0), (101.0, 50.2, 100.3), (0.0, 1.0, 0.0), 40.0, 1.0, 0.0, 1.0))
for w in (1, 2, 4, 8, 16, 64):
    best = None
    for rep in range(5):
        rgb, irgb, rays, ms = rt.render(rt.params(w, 1, 1, seed=rep, max_depth=64))
        trips = rays / w
        if best is None or ms < best[0]:
            best = (ms, rays, trips)
    ms, rays, trips = best
    print(f"{w:2d} live lanes: kernel {ms * 1e3:8.1f} us, {rays} rays, mean path {trips:.1f} queries -> at most {ms * 1e3 / max(trips, 1):.2f} us per trip")
