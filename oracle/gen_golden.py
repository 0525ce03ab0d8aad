#!/usr/bin/env python3
"""Regenerates tests/golden/ from the reference built by oracle/Makefile (needs /root/reference).

    make -C oracle && python oracle/gen_golden.py

Unit vectors come straight from `ref_harness units|scene|paths|dof`; `ref_harness mainppm` runs the reference's
unmodified main() once and its output.ppm is stored gzipped.  Images are rendered by the
reference's own Color()/Camera/Sphere code in row bands (one process per band) and stored as
float32 .npy (linear mean RGB, rows top to bottom).  Two kinds:
  * keyed ("philox"): every uniform comes from the Philox stream keyed on (pixel, sample, bounce);
    the HIP renderer and the C oracle must reproduce these images bit for bit;
  * libc: the loop exactly as main.cpp runs it (rand() stream, unseedable lens generator); high
    sample counts, used for the statistical image test.
"""
import gzip
import json
import os
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
H = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
G = os.path.join(ROOT, "tests", "golden")


def render(scene, nx, ny, ns, mode, seed, aperture, bands):
    edges = np.linspace(0, ny, bands + 1).astype(int)
    tmp = tempfile.mkdtemp()

    def one(b):
        j0, j1 = int(edges[b]), int(edges[b + 1])
        out = os.path.join(tmp, f"b{b}.f32")
        # libc bands get distinct seeds (seed 0 = continue main()'s stream, only valid unbanded)
        s = seed if (mode == "philox" or bands == 1) else seed + b
        r = subprocess.run([H, "render", scene, str(nx), str(ny), str(ns), mode, str(s), str(j0), str(j1), str(aperture), out],
                           check=True, capture_output=True, text=True)
        words = r.stdout.split()               # "rays R hits H"
        return np.fromfile(out, dtype=np.float32).reshape(j1 - j0, nx, 3), (int(words[1]), int(words[3]))

    with ThreadPoolExecutor(max_workers=8) as ex:
        parts = list(ex.map(one, range(bands)))
    # band b covers rows j in [j0, j1) counted from the bottom; images are stored top to bottom
    img = np.concatenate([p[0] for p in reversed(parts)], axis=0)
    return img, sum(p[1][0] for p in parts), sum(p[1][1] for p in parts)


def main():
    # --counts: leave the vectors and images alone; re-render the reproducible images (keyed stream, or
    # aperture 0 so that the unseedable lens generator is multiplied by zero), check that they equal
    # the stored ones bit for bit and refresh only the counters in images.json.
    counts_only = "--counts" in sys.argv[1:]
    os.makedirs(G, exist_ok=True)
    meta = {}
    if counts_only:
        with open(os.path.join(G, "images.json")) as f:
            meta = json.load(f)
    else:
        subprocess.run([H, "scene", os.path.join(G, "scene_final.txt")], check=True)
        subprocess.run([H, "units", G], check=True)
        subprocess.run([H, "paths", os.path.join(G, "color_paths.txt"), "400"], check=True)
        # Camera::GetRay with the lens in use (std::random_device pinned by the harness), and the file the reference's own
        # main() writes: 400x300, 5 samples, about 20 s on one core
        subprocess.run([H, "dof", os.path.join(G, "camera_dof.txt")], check=True)
        with tempfile.TemporaryDirectory() as d:
            subprocess.run([H, "mainppm", d], check=True)
            with open(os.path.join(d, "output.ppm"), "rb") as f:
                text = f.read()
        with open(os.path.join(G, "reference_main_output_400x300x5.ppm.gz"), "wb") as f:
            f.write(gzip.compress(text, 9, mtime=0))
    subprocess.run([H, "tiles", os.path.join(G, "tiles.txt")], check=True)
    with open(os.path.join(G, "stats_report.txt"), "w") as f:
        f.write(subprocess.run([H, "stats"], check=True, capture_output=True, text=True).stdout)
    jobs = [
        # name, scene, nx, ny, ns, mode, seed, aperture(-1 = scene default), bands
        ("libc_final_48x32x3_ap0", "final", 48, 32, 3, "libc", 0, 0.0, 1),
        ("philox_final_96x64x4", "final", 96, 64, 4, "philox", 7, -1, 8),
        ("philox_final_40x30x16_ap0", "final", 40, 30, 16, "philox", 123456789012345, 0.0, 8),
        ("philox_three_hollow_80x45x8", "three_hollow", 80, 45, 8, "philox", 3, -1, 8),
        ("philox_three_64x36x32", "three", 64, 36, 32, "philox", 0, -1, 8),
        ("hispp_final_120x80x512", "final", 120, 80, 512, "libc", 100, -1, 8),
        ("hispp_three_100x56x1024", "three", 100, 56, 1024, "libc", 200, -1, 8),
    ]
    for name, scene, nx, ny, ns, mode, seed, ap, bands in jobs:
        reproducible = mode == "philox" or (ap == 0.0 and bands == 1)
        if counts_only and not reproducible:
            continue
        img, rays, hits = render(scene, nx, ny, ns, mode, seed, ap, bands)
        if counts_only:
            assert np.array_equal(img.view(np.uint32), np.load(os.path.join(G, f"img_{name}.npy")).view(np.uint32)), name
        else:
            np.save(os.path.join(G, f"img_{name}.npy"), img)
        # RenderStats semantics (main_modern.cpp:23-61): every Color() call beyond a sample's first one
        # follows a successful Scatter below the depth cap
        meta[name] = dict(scene=scene, nx=nx, ny=ny, ns=ns, mode=mode, seed=seed, aperture=ap, bands=bands, rays=rays,
                          hits=hits, scatters=rays - nx * ny * ns,
                          mean_rgb=[float(x) for x in img.reshape(-1, 3).mean(axis=0)])
        print(name, rays, meta[name]["mean_rgb"], flush=True)
    with open(os.path.join(G, "images.json"), "w") as f:
        json.dump(meta, f, indent=1)


if __name__ == "__main__":
    sys.exit(main())
