#!/usr/bin/env python3
"""TEST INFRASTRUCTURE -- how far the reference built with its own Release flags (-O3 -mavx2 -mfma: GCC contracts a*b+c into
fused multiply-adds) is from the pinned build without contraction that the golden images and every bit-for-bit claim refer to.
Renders the keyed golden images with oracle/_ref/ref_harness_release and counts differing pixels.

    make -C oracle && python oracle/release_gap.py
"""
import json
import os
import subprocess
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")
H = os.path.join(ROOT, "oracle", "_ref", "ref_harness_release")


def main():
    meta = json.load(open(os.path.join(G, "images.json")))
    for name, m in meta.items():
        if m["mode"] != "philox":
            continue
        with tempfile.TemporaryDirectory() as d:
            out = os.path.join(d, "o.f32")
            r = subprocess.run([H, "render", m["scene"], str(m["nx"]), str(m["ny"]), str(m["ns"]), "philox", str(m["seed"]), "0", str(m["ny"]),
                                str(m["aperture"]), out], capture_output=True, text=True, check=True)
            img = np.fromfile(out, dtype=np.float32).reshape(m["ny"], m["nx"], 3)
        want = np.load(os.path.join(G, f"img_{name}.npy"))
        bits = (img.view(np.uint32) != want.view(np.uint32)).any(axis=2)
        q = lambda a: (255.99 * np.sqrt(a.astype(np.float64))).astype(np.int64)
        print(f"{name}: rays {r.stdout.split()[1]} (pinned build {m['rays']}); pixels differing in any bit {int(bits.sum())} of {bits.size}; "
              f"by more than 1e-3 {int((np.abs(img - want) > 1e-3).any(axis=2).sum())}; quantised integers differ in {int((q(img) != q(want)).any(axis=2).sum())}")


if __name__ == "__main__":
    main()
