"""The C++ host side: classes mirroring the reference surface, the flatten pass and the drop-in main."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "bin")


def _built(name):
    path = os.path.join(BIN, name)
    if not os.path.exists(path):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "raytracing_amd", "host")])
    return path


def test_host_classes_match_reference_vectors():
    """Vec3/Ray/Camera/Sphere/materials/RandomScene written for this repo vs vectors from the reference's classes."""
    out = subprocess.run([_built("host_selftest"), os.path.join(ROOT, "tests", "golden")], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.split("\n")
    assert "ok scene 487" in lines and "ok sphere_hit 1536" in lines and "ok scatter 600" in lines
    assert any(l.startswith("ok camera") for l in lines)
    assert "ok tiles 6" in lines
    # RenderStats::print_stats: same text as the reference's performance.h printed for the same numbers
    report = out.stdout.split("stats report follows\n")[1]
    assert report == open(os.path.join(ROOT, "tests", "golden", "stats_report.txt")).read()


@pytest.mark.gpu
def test_drop_in_main_writes_the_same_ppm_as_the_python_path(tmp_path):
    import raytracing_amd as R
    path = str(tmp_path / "output.ppm")
    out = subprocess.run([_built("raytracing_amd"), "96", "64", "3", path, "5"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert "Rays per second" in out.stdout
    rt = R.Renderer(0)
    rt.upload_scene(R.final_scene_table())
    rt.set_camera(R.final_camera(96, 64))
    _, irgb, rays, _ = rt.render(rt.params(96, 64, 3, seed=5))
    rt.close()
    assert open(path, "rb").read() == R.format_ppm(irgb)
    assert f"Rays traced: {rays}" in out.stdout
    # the same image over every visible GPU of the node (rt_render_multi): the file does not change
    path_all = str(tmp_path / "output_all_gpus.ppm")
    multi = subprocess.run([_built("raytracing_amd"), "96", "64", "3", path_all, "5", "0"], capture_output=True, text=True, timeout=120)
    assert multi.returncode == 0, multi.stderr
    assert open(path_all, "rb").read() == open(path, "rb").read() and f"Rays traced: {rays}" in multi.stdout
    too_many = subprocess.run([_built("raytracing_amd"), "96", "64", "3", path_all, "5", "99"], capture_output=True, text=True, timeout=120)
    assert too_many.returncode != 0 and "visible" in too_many.stderr
    # error path: unwritable output
    bad = subprocess.run([_built("raytracing_amd"), "8", "8", "1", "/nonexistent_dir/x.ppm"], capture_output=True, text=True, timeout=120)
    assert bad.returncode != 0


@pytest.mark.gpu
def test_modern_main_reports_reference_statistics_and_renders_progressively(tmp_path):
    """main_modern.cpp's drop-in: band-by-band rendering gives the single-call image and the RenderStats counters."""
    import raytracing_amd as R
    path = str(tmp_path / "modern.ppm")
    out = subprocess.run([_built("raytracing_modern_amd"), "100", "70", "4", path, "9", "16"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    rt = R.Renderer(0)
    rt.upload_scene(R.final_scene_table())
    rt.set_camera(R.final_camera(100, 70))
    _, irgb, rays, _ = rt.render(rt.params(100, 70, 4, seed=9))
    c = rt.counters()
    rt.close()
    assert open(path, "rb").read() == R.format_ppm(irgb)
    assert c[0] == rays
    for line in ("=== Modern Raytracer ===", "Resolution: 100x70", "Samples per pixel: 4", "Rendering complete!",
                 f"Rays traced: {c[0]}", f"Intersection tests: {c[1]}", f"Material evaluations: {c[2]}"):
        assert line in out.stdout, line
    assert "tiles 35 / 35" in out.stderr
    # one call for the whole image: same file
    path2 = str(tmp_path / "modern_one_call.ppm")
    one = subprocess.run([_built("raytracing_modern_amd"), "100", "70", "4", path2, "9", "0"], capture_output=True, text=True, timeout=120)
    assert one.returncode == 0, one.stderr
    assert open(path2, "rb").read() == open(path, "rb").read()
