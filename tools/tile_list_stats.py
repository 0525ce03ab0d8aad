#!/usr/bin/env python3
"""Lengths of the per-tile sphere lists of camera rays (rt_test_tile_lists) for the benchmark scenes."""
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracing_amd as R

rt = R.Renderer(0)
for name, table, (W, H) in (("final scene 1200x800", R.final_scene_table(), (1200, 800)), ("final scene 3840x2160", R.final_scene_table(), (3840, 2160)),
                            ("20 000-sphere grid 1920x1080", R.scene_grid(42, 75, 19996), (1920, 1080)), ("three spheres 400x225", R.three_sphere_table(), (400, 225))):
    rt.upload_scene(table)
    rt.set_camera(R.final_camera(W, H) if len(table) > 10 else R.three_camera(W, H))
    counts, _ = rt.test_tile_lists(rt.params(W, H, 1))
    have = counts[counts >= 0]
    print(f"{name}: {len(counts)} tiles, {100.0 * (counts < 0).mean():.1f} % without a list; lists: mean {have.mean():.2f}, median {np.median(have):.0f}, p90 {np.percentile(have, 90):.0f}, max {have.max()}")
