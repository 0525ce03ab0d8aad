#!/usr/bin/env python3
"""Soak run: many seeded random scenes and cameras (near, far, inside spheres), every kernel organisation against the
literal one-lane-per-pixel kernel, bit for bit.  GPU only (no oracle), so it covers far more cases than the test suite."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import raytracing_amd as R

def random_scene(rng, n):
    t = np.zeros((n, 9))
    spread = rng.choice([1.0, 3.0, 12.0, 80.0])
    t[:, 0:3] = rng.uniform(-spread, spread, size=(n, 3))
    t[:, 3] = np.exp(rng.uniform(np.log(0.01), np.log(0.6 * spread), size=n))
    if rng.random() < 0.4:        # (nearly) equal radii: whole chunks of 32 then take the filter's normalised form
        t[:, 3] = float(np.exp(rng.uniform(np.log(0.02), np.log(0.3 * spread)))) * rng.uniform(rng.choice([0.9, 0.999, 1.0]), 1.0, size=n)
    t[:, 4] = rng.integers(0, 3, size=n)
    t[:, 5:8] = rng.uniform(0.05, 1.0, size=(n, 3))
    for k in range(n):
        if t[k, 4] == R.METAL: t[k, 8] = rng.choice([0.0, 0.3, 1.0])
        elif t[k, 4] == R.DIELECTRIC:
            t[k, 8] = rng.choice([1.5, 0.8, 1.0, 2.4]); t[k, 5:8] = 1.0
            if rng.random() < 0.4: t[k, 3] = -t[k, 3]
        if k > 0 and rng.random() < 0.1: t[k, 0:4] = t[rng.integers(0, k), 0:4]
        if rng.random() < 0.03: t[k, 3] = rng.choice([0.0, 1e-9, 1e-6, 2e-6, 4e-6, 5e5, 6e5, 1e7])     # outside what the plane filter judges itself
    if n > 3 and rng.random() < 0.5: t[0] = [0.0, -1000.0, 0.0, 1000.0, R.LAMBERTIAN, 0.5, 0.5, 0.5, 0.0]
    return np.asarray(t, dtype=np.float32).astype(np.float64), spread

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
rt = R.Renderer(0, camera_ray_lists=True)
swept = R.Renderer(0, camera_ray_lists=False)          # camera rays through the brute-force sweep like every other ray
W, H, S = 64, 40, 3
t0 = time.time(); bad = 0
for case in range(cases):
    n = int(rng.choice([1, 3, 7, 33, 100, 487, 600, 1537, 2500, 5000]))
    table, spread = random_scene(rng, n)
    dist = float(rng.choice([0.5, 2.0, 6.0, 50.0, 2000.0])) * spread
    direction = rng.normal(size=3); direction /= np.linalg.norm(direction)
    look_at = rng.uniform(-0.3, 0.3, size=3) * spread
    vfov = float(np.clip(np.degrees(2 * np.arctan(1.5 * spread / max(dist, 1e-3))), 0.01, 120.0))
    cam = R.Camera(look_at + direction * dist, look_at, (0.0, 1.0, 0.0), vfov, np.float32(W) / np.float32(H), float(rng.choice([0.0, 0.2, 0.2, 2.0]) * (spread if rng.random() < 0.3 else 1.0)), max(dist * float(rng.choice([1.0, 1.0, 0.3, 3.0])), 0.1))
    rt.upload_scene(table); rt.set_camera(cam)
    seed = 5000 + case
    lit = rt.render(rt.params(W, H, S, seed=seed, mode=R.MODE_PIXEL))
    swept.upload_scene(table); swept.set_camera(cam)
    for mode, name in ((R.MODE_MEGAKERNEL, "megakernel"), (-1, "megakernel, camera rays swept"), (R.MODE_BVH, "bvh"), (R.MODE_WAVEFRONT, "wavefront")):
        got = swept.render(swept.params(W, H, S, seed=seed, mode=R.MODE_MEGAKERNEL)) if mode < 0 else rt.render(rt.params(W, H, S, seed=seed, mode=mode))
        a, b = got[0].view(np.uint32), lit[0].view(np.uint32)
        same = ((got[0] == lit[0]) | (np.isnan(got[0]) & np.isnan(lit[0])))
        if got[2] != lit[2] or not same.all():
            bad += 1
            print("MISMATCH case %d n %d spread %g dist %g mode %s: rays %d vs %d, %d differing values" % (case, n, spread, dist, name, got[2], lit[2], int((~same).sum())), flush=True)
    if case % 25 == 24: print("case %d done, %.0f s, mismatches %d" % (case + 1, time.time() - t0, bad), flush=True)
print("soak finished: %d cases, %d mismatches" % (cases, bad))
sys.exit(1 if bad else 0)
