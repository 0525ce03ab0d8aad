"""Scene tables and cameras of the benchmark configurations (host-side setup, untimed).

A scene table has one row per sphere: cx cy cz radius material albedo_r albedo_g albedo_b param,
with material 0 = Lambertian, 1 = Metal, 2 = Dielectric and param = fuzz | refraction index.
"""
import os

import numpy as np

from ._lib import Camera

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")


def final_scene_table():
    """The 487 spheres RandomScene() builds (main.cpp:41-83) from glibc's default rand() stream, as
    flattened from the reference's own objects; stored because the constructor-argument evaluation
    order that decides which draw lands where is compiler-specific."""
    return np.loadtxt(os.path.join(_DATA, "scene_final.txt"), comments="#", dtype=np.float64, ndmin=2)


def three_sphere_table(hollow=False):
    """Ground + diffuse + glass + fuzzy metal (performance_comparison.cpp:48-51); `hollow` adds the
    negative-radius glass shell of main.cpp:107."""
    rows = [[0.0, -100.5, -1.0, 100.0, 0, 0.5, 0.5, 0.5, 0.0],
            [0.0, 0.0, -1.0, 0.5, 0, 0.7, 0.3, 0.3, 0.0],
            [-1.0, 0.0, -1.0, 0.5, 2, 1.0, 1.0, 1.0, 1.5],
            [1.0, 0.0, -1.0, 0.5, 1, 0.8, 0.6, 0.2, 0.3]]
    if hollow:
        rows.append([-1.0, 0.0, -1.0, -0.45, 2, 1.0, 1.0, 1.0, 1.5])
    return np.array(rows, dtype=np.float64)


def make_camera(look_from, look_at, up, vfov, aspect, aperture, focus_dist):
    return Camera(look_from, look_at, up, vfov, aspect, aperture, focus_dist)


def final_camera(width, height, aperture=0.1):
    """main.cpp:111-116."""
    return Camera((13.0, 2.0, 3.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 20.0, np.float32(width) / np.float32(height), aperture, 10.0)


def three_camera(width, height):
    """performance_comparison.cpp:66-72."""
    return Camera((0.0, 0.0, 0.0), (0.0, 0.0, -1.0), (0.0, 1.0, 0.0), 90.0, np.float32(width) / np.float32(height), 0.0, 1.0)
