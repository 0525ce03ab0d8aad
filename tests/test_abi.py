"""The C-ABI library loads without a GPU and exports every symbol include/rt_amd.h declares."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import raytracing_amd as R
from raytracing_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "rt_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rt_[a-z_0-9]+)\s*\(", text)))


def test_every_declared_symbol_is_exported():
    L = R.load_library()
    names = _declared()
    assert len(names) >= 15
    for n in names:
        assert hasattr(L, n), n
    assert sorted(_lib.EXPORTS) == names


def test_struct_layouts_match_header():
    assert C.sizeof(R.CameraPod) == 22 * 4
    assert C.sizeof(R.Params) == 48 and R.Params.seed.offset == 24 and R.Params.mode.offset == 44


def test_owned_rows_banding():
    L = R.load_library()
    P = R.Renderer.params
    assert L.rt_owned_rows(C.byref(P(10, 800, 1))) == 800
    # interleaved 8-row bands over 8 ranks cover every row exactly once
    assert sum(L.rt_owned_rows(C.byref(P(10, 2160, 1, band_rows=8, band_first=r, band_step=8))) for r in range(8)) == 2160
    # ragged last band
    assert [L.rt_owned_rows(C.byref(P(10, 50, 1, band_rows=16, band_first=r, band_step=2))) for r in range(2)] == [32, 18]
    assert L.rt_owned_rows(C.byref(P(10, 0, 1))) == 0
    assert L.rt_owned_pixels(C.byref(P(10, 50, 1, band_rows=16, band_first=1, band_step=2))) == 180


def test_ppm_bytes_match_reference_format():
    """main.cpp:95-97,143-144: "P3\\n{nx} {ny}\\n255\\n" then "r g b\\n" per pixel."""
    irgb = np.array([[[0, 1, 2], [255, 128, 7]], [[12, 34, 56], [-2147483648, 9, 100]]], dtype=np.int32)
    want = "P3\n2 2\n255\n" + "".join(f"{p[0]} {p[1]} {p[2]}\n" for row in irgb for p in row)
    assert R.format_ppm(irgb) == want.encode()
    path = os.path.join(ROOT, "build", "test_out.ppm")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    R.write_ppm(path, irgb)
    assert open(path, "rb").read() == want.encode()
    # larger random images, every value the quantiser can produce, against Python's own formatting
    rng = np.random.default_rng(3)
    for (w, h) in [(1, 1), (17, 9), (301, 77)]:
        img = rng.integers(0, 256, size=(h, w, 3), dtype=np.int32)
        img.reshape(-1)[rng.integers(0, img.size, size=min(8, img.size))] = [-1, 256, 1000, -2147483648, 2147483647, 99999, 10, 100][:min(8, img.size)]
        text = f"P3\n{w} {h}\n255\n" + "".join(f"{p[0]} {p[1]} {p[2]}\n" for row in img for p in row)
        assert R.format_ppm(img) == text.encode()
    # binary P6 form: header + one byte per channel; values outside a byte are refused
    img = rng.integers(0, 256, size=(9, 17, 3), dtype=np.int32)
    R.write_ppm_binary(path, img)
    assert open(path, "rb").read() == b"P6\n17 9\n255\n" + img.astype(np.uint8).tobytes()
    img[2, 3, 1] = 256
    with pytest.raises(R.RtError):
        R.write_ppm_binary(path, img)


def test_null_and_invalid_arguments_are_reported_not_fatal():
    L = R.load_library()
    assert L.rt_device_count(None) == -1
    assert L.rt_create(0, None) == -1
    assert L.rt_upload_scene(None, 1, None, None, None, None, None, None, None) == -1
    assert L.rt_render_stats(None, None, None) == -1
    assert b"null" in L.rt_last_error(None)
    L.rt_destroy(None)


def test_scene_grid_is_deterministic_and_follows_the_reference_rule():
    """The 20 000-sphere stress scene: RandomScene()'s layout rule (main.cpp:41-83) on a 150 x 150 grid."""
    t = R.scene_grid(42, 75, 19996)
    assert t.shape == (20000, 9)
    assert np.array_equal(t, R.scene_grid(42, 75, 19996))
    assert not np.array_equal(t, R.scene_grid(43, 75, 19996))
    assert tuple(t[0, :5]) == (0.0, -1000.0, 0.0, 1000.0, 0.0)
    small = t[1:-3]
    assert np.all(small[:, 3] == np.float32(0.2)) and np.all(small[:, 1] == np.float32(0.2))
    frac = [np.mean(small[:, 4] == m) for m in (0, 1, 2)]
    assert abs(frac[0] - 0.80) < 0.01 and abs(frac[1] - 0.15) < 0.01 and abs(frac[2] - 0.05) < 0.01
    d = np.sqrt((small[:, 0] - 4.0) ** 2 + small[:, 2] ** 2)
    assert d.min() > 0.9                                   # exclusion zone around (4, 0.2, 0)
    cell = np.floor(small[:, [0, 2]])
    jitter = small[:, [0, 2]] - cell
    assert jitter.min() >= 0.0 and jitter.max() < 0.9 + 1e-6
    assert np.all(small[small[:, 4] == 1, 8] < 0.5) and np.all(small[small[:, 4] == 2, 8] == 1.5)
    assert [tuple(r[:5]) for r in t[-3:]] == [(0, 1, 0, 1, 2), (-4, 1, 0, 1, 0), (4, 1, 0, 1, 1)]
    # the reference-sized layout: 22 x 22 cells
    assert 470 <= len(R.scene_grid(1, 11, 10 ** 6)) <= 488


def test_sphere_hierarchy_invariants():
    """The tree RT_MODE_BVH walks: every sphere in exactly one leaf or in the direct list, boxes hold the centres
    below them, rmax is the largest radius below, children adjacent, depth within the traversal stack."""
    from raytracing_amd._lib import build_hierarchy
    rng = np.random.default_rng(5)
    tables = [R.final_scene_table(), R.three_sphere_table(True), R.scene_grid(42, 75, 19996),
              np.array([[0, 0, -1, 0.5, 0, 0.5, 0.5, 0.5, 0]], dtype=np.float64)]
    dup = np.tile(np.array([[1.0, 2.0, 3.0, 0.25, 0, 0.5, 0.5, 0.5, 0.0]]), (37, 1))     # 37 identical spheres
    tables.append(dup)
    for table in tables:
        n = len(table)
        nodes, leaf, big, depth = build_hierarchy(table)
        assert depth + 2 <= 24
        link = nodes[:, 7].copy().view(np.int32)
        seen = np.zeros(n, dtype=int)
        seen[big] += 1
        centres, radii = table[:, 0:3].astype(np.float32), np.abs(table[:, 3].astype(np.float32))

        def below(i):
            if link[i] < 0:
                code = -(int(link[i]) + 1)
                first, count = code >> 3, (code & 7) + 1
                ids = leaf[first:first + count]
                assert list(ids) == sorted(ids)
                return list(ids)
            l = int(link[i])
            return below(l) + below(l + 1)

        import sys
        sys.setrecursionlimit(10000)
        if len(big) < n:
            for i in range(len(nodes)):
                ids = below(i) if (i == 0 or True) else []
                c = centres[ids]
                assert (c >= nodes[i, 0:3]).all() and (c <= nodes[i, 4:7]).all()
                rmax = radii[ids].max()
                assert rmax <= nodes[i, 3] <= rmax + 1e-5 * (1.0 + np.abs(nodes[i, 0:3]).sum() + np.abs(nodes[i, 4:7]).sum())
                if i == 0:
                    seen[ids] += 1
        assert (seen == 1).all()
    ground = R.final_scene_table()
    _, _, big, _ = build_hierarchy(ground)
    assert 0 in big                      # the r = 1000 ground sphere is tested directly, not boxed
    # spheres with a non-finite centre or radius (no ray hits them in the reference's arithmetic) stay out of the hierarchy
    # and leave it well formed
    odd = R.final_scene_table().copy()
    bad = [3, 40, 41, 200, 486]
    odd[3, 0] = np.nan; odd[40, 2] = np.inf; odd[41, 1] = -np.inf; odd[200, 3] = np.nan; odd[486, 3] = np.inf
    nodes, leaf, big, depth = build_hierarchy(odd)
    listed = sorted(list(leaf) + list(big))
    assert listed == [i for i in range(len(odd)) if i not in bad]
    assert np.isfinite(nodes[:, 0:7]).all() and depth + 2 <= 24
    nodes, leaf, big, depth = build_hierarchy(np.array([[np.nan, 0, 0, 1.0, 0, 0.5, 0.5, 0.5, 0]]))      # nothing usable at all
    assert len(big) == 0


def test_format_ppm_reproduces_the_file_the_reference_wrote():
    """output.ppm written by the reference's own main() (main.cpp:91-97, 143-147; generated once by
    `oracle/_ref/ref_harness mainppm`): rt_format_ppm of its integers gives back the file byte for byte."""
    import gzip
    with gzip.open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_main_output_400x300x5.ppm.gz"), "rb") as f:
        text = f.read()
    tok = text.split()
    assert tok[:4] == [b"P3", b"400", b"300", b"255"]
    img = np.array(tok[4:], dtype=np.int32).reshape(300, 400, 3)
    assert R.format_ppm(img) == text
