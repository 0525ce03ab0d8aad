"""HIP render path against the reference's golden vectors and against the CPU oracle.

All calls go through the C ABI (librt_amd.so).  Integer outputs and, by construction of the exact
code paths, float outputs are compared bit for bit; the tolerance the task states for float
results (1e-5 relative) is therefore met with margin and is only invoked where a test says so.
"""
import ctypes as C
import json
import os
import sys

import numpy as np
import pytest

import _oracle as O
import raytracing_amd as R

pytestmark = pytest.mark.gpu
G = O.GOLDEN


@pytest.fixture(scope="module", params=["swept", "lists"])
def rt(request):
    """Every test runs on two contexts: one whose camera rays go through the brute-force sweep like every other ray (the path
    the benchmark times), one whose camera rays use the per-tile sphere lists (the library's default)."""
    r = R.Renderer(0, camera_ray_lists=(request.param == "lists"))
    yield r
    r.close()


def _rows(name):
    out = []
    with open(os.path.join(G, name)) as f:
        for line in f:
            if not line.startswith("#") and line.strip():
                out.append(line.split())
    return out


def same(a, b):
    a = np.asarray(a, dtype=np.float32)
    b = np.asarray(b, dtype=np.float32)
    return bool(np.all((a == b) | (np.isnan(a) & np.isnan(b))))


def mismatch(a, b):
    a = np.asarray(a, dtype=np.float32)
    b = np.asarray(b, dtype=np.float32)
    return int(np.sum(~((a == b) | (np.isnan(a) & np.isnan(b)))))


def test_philox_known_answers(rt):
    ctr = [[0, 0, 0, 0]]
    assert tuple(rt.test_philox(ctr, [0, 0])[0]) == (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)
    assert tuple(rt.test_philox([[0xffffffff] * 4], [0xffffffff] * 2)[0]) == (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)
    assert tuple(rt.test_philox([[0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344]], [0xa4093822, 0x299f31d0])[0]) == \
        (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)
    # against the oracle on a batch of counters
    rng = np.random.default_rng(1)
    c = rng.integers(0, 2 ** 32, size=(257, 4), dtype=np.uint64).astype(np.uint32)
    got = rt.test_philox(c, [123, 456])
    for e in range(len(c)):
        cc = (C.c_uint32 * 4)(*[int(x) for x in c[e]])
        kk = (C.c_uint32 * 2)(123, 456)
        oo = (C.c_uint32 * 4)()
        O.lib().orc_philox4x32_10(cc, kk, oo)
        assert tuple(got[e]) == tuple(oo)


def test_sphere_hit_vs_reference_vectors(rt):
    t = np.array([[np.float32(float(x)) for x in r] for r in _rows("sphere_hit.txt")], dtype=np.float32)
    hit, tt, p, n = rt.test_sphere_hit(t[:, 0:3], t[:, 3:6], t[:, 6:9], t[:, 9], t[:, 10], t[:, 11])
    assert np.array_equal(hit, t[:, 12].astype(np.int32))
    assert same(tt, t[:, 13]) and same(p, t[:, 14:17]) and same(n, t[:, 17:20])
    # front_face as Dielectric::Scatter derives it (material.h:148): dot(d, n) <= 0
    h = hit == 1
    ff_gpu = np.einsum("ij,ij->i", t[h, 3:6].astype(np.float64), n[h].astype(np.float64)) <= 0
    ff_ref = np.einsum("ij,ij->i", t[h, 3:6].astype(np.float64), t[h, 17:20].astype(np.float64)) <= 0
    assert np.array_equal(ff_gpu, ff_ref)


def test_scatter_vs_reference_vectors(rt):
    rows = _rows("scatter.txt")
    typ = np.array([int(r[0]) for r in rows], np.int32)
    v = np.array([[np.float32(float(x)) for x in r[1:18]] for r in rows], np.float32)
    draws = np.array([[int(x) for x in r[18:66]] for r in rows], np.int32)
    tail = [r[66:] for r in rows]
    want_ret = np.array([int(x[0]) for x in tail], np.int32)
    want = np.array([[np.float32(float(y)) for y in x[1:10]] for x in tail], np.float32)
    want_used = np.array([int(x[10]) for x in tail], np.int32)
    param = np.array([np.float32(float(x[11])) if t == 1 else p for x, t, p in zip(tail, typ, v[:, 3])], np.float32)
    ret, att, oo, od, used = rt.test_scatter(typ, v[:, 0:3], param, v[:, 4:7], v[:, 7:10], v[:, 11:14], v[:, 14:17], draws)
    assert np.array_equal(ret, want_ret)
    assert np.array_equal(used, want_used)
    assert same(att, want[:, 0:3]) and same(oo, want[:, 3:6])
    # scatter directions: the stated bar is 1e-5 relative; the exact path gives equality
    np.testing.assert_allclose(od, want[:, 6:9], rtol=1e-5, atol=1e-7)
    assert same(od, want[:, 6:9])


def test_get_ray_vs_reference_vectors(rt):
    for r in _rows("camera.txt"):
        if r[0] != "ray":
            continue
        v = [np.float32(float(x)) for x in r[1:]]
        if v[11] != 0.0:
            continue    # rows with the lens in use are in camera_dof.txt, which also holds the disk sample of each
        cam = R.make_camera(v[0:3], v[3:6], v[6:9], v[9], v[10], v[11], v[12])
        ro, rd = rt.test_get_ray(cam, [v[13]], [v[14]], [[0.0, 0.0]])
        assert same(ro[0], v[15:18]) and same(rd[0], v[18:21])
    # defocus disk (camera.cpp:31-53): rays returned by the reference's own Camera::GetRay with std::random_device pinned,
    # each with the unit-disk sample its lens generator drew -- bit for bit
    rows = [[np.float32(float(x)) for x in r[1:]] for r in _rows("camera_dof.txt")]
    assert len(rows) >= 200
    by_cam = {}
    for v in rows:
        by_cam.setdefault(tuple(v[0:13]), []).append(v)
    for key, group in by_cam.items():
        cam = R.make_camera(key[0:3], key[3:6], key[6:9], key[9], key[10], key[11], key[12])
        g = np.array(group, dtype=np.float32)
        ro, rd = rt.test_get_ray(cam, g[:, 13], g[:, 14], g[:, 15:17])
        assert same(ro, g[:, 18:21]) and same(rd, g[:, 21:24])
        assert not same(ro, np.broadcast_to(np.array(key[0:3], np.float32), ro.shape))     # the lens offset is really there


def test_reflect_refract_schlick_vs_reference_vectors(rt):
    """material.h:7-38 through the GPU's own device functions.  Reflect and Refract are the reference's float operations
    in the reference's order; Schlick's pow(1 - cosine, 5), a libm call in double in the reference, is three double products
    on the GPU: the number of vectors on which the narrowed float differs is reported and has to be zero here."""
    t = np.array([[np.float32(float(x)) for x in r] for r in _rows("helpers.txt")], dtype=np.float32)
    refl, ok, refr, sch = rt.test_helpers(t[:, 0:3], t[:, 3:6], t[:, 6], t[:, 7], t[:, 8])
    assert same(refl, t[:, 9:12])
    assert np.array_equal(ok, t[:, 12].astype(np.int32)) and 50 < ok.sum() < len(t)
    assert same(refr, t[:, 13:16])
    differing = mismatch(sch, t[:, 16])
    print(f"Schlick: {differing} of {len(t)} values differ from the reference's libm pow")
    assert differing == 0


def _meta():
    with open(os.path.join(G, "images.json")) as f:
        return json.load(f)


def _setup(rt, m):
    if m["scene"] == "final":
        ap = 0.1 if m["aperture"] < 0 else m["aperture"]
        rt.upload_scene(R.final_scene_table())
        rt.set_camera(R.final_camera(m["nx"], m["ny"], ap))
    else:
        rt.upload_scene(R.three_sphere_table(m["scene"] == "three_hollow"))
        rt.set_camera(R.three_camera(m["nx"], m["ny"]))


@pytest.mark.parametrize("mode", [R.MODE_MEGAKERNEL, R.MODE_WAVEFRONT, R.MODE_PIXEL, R.MODE_BVH])
@pytest.mark.parametrize("name", ["philox_final_96x64x4", "philox_final_40x30x16_ap0",
                                  "philox_three_hollow_80x45x8", "philox_three_64x36x32"])
def test_render_is_bit_identical_to_reference_image(rt, name, mode):
    """Images written by the reference's own Color()/Sphere/Material code on the keyed stream."""
    m = _meta()[name]
    want = np.load(os.path.join(G, f"img_{name}.npy"))
    _setup(rt, m)
    rgb, irgb, rays, ms = rt.render(rt.params(m["nx"], m["ny"], m["ns"], seed=m["seed"], mode=mode))
    assert rays == m["rays"]
    # RenderStats of main_modern.cpp:23-61 as counted by the reference build
    assert rt.counters() == (m["rays"], m["hits"], m["scatters"])
    assert mismatch(rgb, want) == 0
    q = np.zeros_like(irgb)
    for y in range(rgb.shape[0]):
        for x in range(rgb.shape[1]):
            O.lib().orc_quantise(rgb[y, x].ctypes.data_as(C.POINTER(C.c_float)), q[y, x].ctypes.data_as(C.POINTER(C.c_int)))
    assert np.array_equal(irgb, q)


def test_band_split_is_invisible(rt):
    """Interleaved bands rendered separately assemble to the single-call image, bit for bit."""
    rt.upload_scene(R.final_scene_table())
    rt.set_camera(R.final_camera(72, 50))
    whole, iw, rays, _ = rt.render(rt.params(72, 50, 3, seed=11))
    for step, band in ((2, 8), (4, 3), (8, 16)):
        img = np.zeros_like(whole)
        total = 0
        for first in range(step):
            p = rt.params(72, 50, 3, seed=11, band_rows=band, band_first=first, band_step=step)
            part, _, r, _ = rt.render(p)
            total += r
            row = 0
            for b in range(first, (50 + band - 1) // band, step):
                y0, y1 = b * band, min((b + 1) * band, 50)
                img[y0:y1] = part[row:row + (y1 - y0)]
                row += y1 - y0
            assert row == part.shape[0]
        assert same(img, whole) and total == rays


def test_statistical_image_bound_vs_high_spp_reference(rt):
    """Image-level check against the reference's own renders at high sample counts.

    Bound: both images are Monte-Carlo means; per-pixel standard deviation of one sample is
    at most 0.5 (radiance lies in [0,1]) and about 0.25 on average, so the expected mean absolute difference of two
    independent estimates with n1 and n2 samples is below 0.25*sqrt(1/n1 + 1/n2)*0.8.  The
    asserted bounds leave a factor two on that, and PSNR = -10 log10(MSE) follows from it."""
    meta = _meta()
    for name, spp in (("hispp_final_120x80x512", 256), ("hispp_three_100x56x1024", 256)):
        m = meta[name]
        want = np.load(os.path.join(G, f"img_{name}.npy")).astype(np.float64)
        _setup(rt, m)
        rgb, _, _, _ = rt.render(rt.params(m["nx"], m["ny"], spp, seed=99))
        rgb = rgb.astype(np.float64)
        mae = np.abs(rgb - want).mean()
        mse = ((rgb - want) ** 2).mean()
        psnr = -10 * np.log10(mse)
        bound = 2 * 0.25 * 0.8 * np.sqrt(1.0 / spp + 1.0 / m["ns"])
        assert mae < bound, (name, mae, bound)
        assert psnr > 30.0, (name, psnr)
        np.testing.assert_allclose(rgb.reshape(-1, 3).mean(0), want.reshape(-1, 3).mean(0), rtol=0.01)


def test_full_size_rows_match_oracle(rt):
    """Config-1 geometry (1200x800, 10 spp, depth 50): rows of the full image against the oracle."""
    scene = O.SceneArrays.from_table(O.load_scene_table())
    ocam = O.final_camera(1200, 800)
    rt.upload_scene(R.final_scene_table())
    rt.set_camera(R.final_camera(1200, 800))
    # rows y = 600..603 from the top (sphere field) as band 150 of 4-row bands
    p = rt.params(1200, 800, 10, seed=0, band_rows=4, band_first=150, band_step=1000)
    rgb, irgb, rays, _ = rt.render(p)
    assert rgb.shape[0] == 4
    want, wq, wrays = O.render(scene, ocam, 1200, 800, 10, O.usrc_philox(0), 800 - 604, 800 - 600)
    assert rays == wrays and same(rgb, want) and np.array_equal(irgb, wq)


def test_full_size_properties(rt):
    """Config 1 whole image: determinism, band checksum-of-checksums, value range, ray statistics."""
    rt.upload_scene(R.final_scene_table())
    rt.set_camera(R.final_camera(1200, 800))
    a, ia, rays_a, _ = rt.render(rt.params(1200, 800, 10, seed=0))
    b, ib, rays_b, _ = rt.render(rt.params(1200, 800, 10, seed=0))
    assert rays_a == rays_b and np.array_equal(a.view(np.uint32), b.view(np.uint32))
    halves = [rt.render(rt.params(1200, 800, 10, seed=0, band_rows=8, band_first=f, band_step=2)) for f in (0, 1)]
    assert halves[0][2] + halves[1][2] == rays_a
    s = sum(int(h[1].astype(np.int64).sum()) for h in halves)
    assert s == int(ia.astype(np.int64).sum())
    assert np.isfinite(a).all() and a.min() >= 0.0 and a.max() <= 1.0 and ia.min() >= 0 and ia.max() <= 255
    assert 2.4 < rays_a / (1200 * 800 * 10) < 3.0            # the reference traces about 2.7 rays per sample
    np.testing.assert_allclose(a.reshape(-1, 3).mean(0), [0.2995, 0.3498, 0.4462], atol=0.004)
    c, _, _, _ = rt.render(rt.params(1200, 800, 10, seed=1))
    assert not np.array_equal(a, c)
    # the wavefront organisation of the same loop: same image, same ray count
    w, iw, rays_w, _ = rt.render(rt.params(1200, 800, 10, seed=0, mode=R.MODE_WAVEFRONT))
    assert rays_w == rays_a and np.array_equal(w.view(np.uint32), a.view(np.uint32)) and np.array_equal(iw, ia)
    # and the hierarchy walk in place of the brute-force sweep
    h, ih, rays_h, _ = rt.render(rt.params(1200, 800, 10, seed=0, mode=R.MODE_BVH))
    assert rays_h == rays_a and np.array_equal(h.view(np.uint32), a.view(np.uint32)) and np.array_equal(ih, ia)


def test_stress_scene_rows_match_oracle(rt):
    """Config 5 (20 000 spheres, 1920x1080): scene records no longer fit LDS; rows against the oracle."""
    table = R.scene_grid(42, 75, 19996)
    assert len(table) == 20000
    scene = O.SceneArrays.from_table(table)
    ocam = O.final_camera(1920, 1080)
    rt.upload_scene(table)
    rt.set_camera(R.final_camera(1920, 1080))
    # two image rows in the sphere field (y = 700, 701) = band 350 of 2-row bands, 1 sample
    p = rt.params(1920, 1080, 1, seed=4, band_rows=2, band_first=350, band_step=10000)
    rgb, irgb, rays, _ = rt.render(p)
    want, wq, wrays = O.render(scene, ocam, 1920, 1080, 1, O.usrc_philox(4), 1080 - 702, 1080 - 700, fast=False)
    assert rays == wrays and same(rgb, want) and np.array_equal(irgb, wq)
    # the hierarchy walk on the same rows, and on a larger part of the image against the brute-force kernel
    p_bvh = rt.params(1920, 1080, 1, seed=4, band_rows=2, band_first=350, band_step=10000, mode=R.MODE_BVH)
    rgb_b, irgb_b, rays_b, _ = rt.render(p_bvh)
    assert rays_b == wrays and same(rgb_b, want) and np.array_equal(irgb_b, wq)
    brute = rt.render(rt.params(1920, 1080, 2, seed=9, band_rows=40, band_first=3, band_step=9))
    walk = rt.render(rt.params(1920, 1080, 2, seed=9, band_rows=40, band_first=3, band_step=9, mode=R.MODE_BVH))
    assert brute[2] == walk[2] and mismatch(brute[0], walk[0]) == 0 and np.array_equal(brute[1], walk[1])
    # a mid-size scene whose count is not a multiple of the filter's 32-sphere chunk
    table = R.scene_grid(7, 20, 10 ** 6)
    assert len(table) % 32 != 0 and len(table) > 1536
    rt.upload_scene(table)
    rt.set_camera(R.final_camera(96, 64))
    rgb, irgb, rays, _ = rt.render(rt.params(96, 64, 2, seed=8))
    want, wq, wrays = O.render(O.SceneArrays.from_table(table), O.final_camera(96, 64), 96, 64, 2, O.usrc_philox(8))
    assert rays == wrays and same(rgb, want) and np.array_equal(irgb, wq)
    both = [rt.render(rt.params(96, 64, 2, seed=8, mode=m))[0] for m in (R.MODE_MEGAKERNEL, R.MODE_PIXEL, R.MODE_WAVEFRONT, R.MODE_BVH)]
    assert same(both[0], both[1]) and same(both[0], both[2]) and same(both[0], both[3])


def test_large_scene_operand_tiles_in_lds(rt, monkeypatch):
    """Scenes whose records do not fit LDS, with the matrix-pipe filter's operands staged in LDS as tiles of 512 spheres that
    the twelve waves of a block share and sweep in step (RT_AMD_LARGE_SCENES=tiles; the default streams them from L2 per wave):
    config 5's 20 000 spheres against the oracle, scenes around the tile and chunk boundaries against the default form, and
    small scenes forced off the LDS mirror (one tile, two tiles, a last chunk that is not full)."""
    monkeypatch.setenv("RT_AMD_LARGE_SCENES", "tiles")
    tiled = R.Renderer(0, camera_ray_lists=rt.camera_ray_lists)
    table = R.scene_grid(42, 75, 19996)
    tiled.upload_scene(table)
    tiled.set_camera(R.final_camera(1920, 1080))
    p = tiled.params(1920, 1080, 1, seed=4, band_rows=2, band_first=350, band_step=10000)
    rgb, irgb, rays, _ = tiled.render(p)
    want, wq, wrays = O.render(O.SceneArrays.from_table(table), O.final_camera(1920, 1080), 1920, 1080, 1, O.usrc_philox(4), 1080 - 702, 1080 - 700, fast=False)
    assert rays == wrays and same(rgb, want) and np.array_equal(irgb, wq)
    rt.upload_scene(table)
    rt.set_camera(R.final_camera(1920, 1080))
    pb = dict(seed=9, band_rows=40, band_first=3, band_step=9)
    a, b = tiled.render(tiled.params(1920, 1080, 2, **pb)), rt.render(rt.params(1920, 1080, 2, **pb))
    assert a[2] == b[2] and mismatch(a[0], b[0]) == 0 and np.array_equal(a[1], b[1])
    rng = np.random.default_rng(99)
    for n in (522, 544, 545, 1023, 1024, 1025, 1537, 4097):
        table = _random_scene(rng, n)
        for r in (tiled, rt):
            r.upload_scene(table)
            r.set_camera(R.final_camera(72, 40))
        a, b = tiled.render(tiled.params(72, 40, 3, seed=n)), rt.render(rt.params(72, 40, 3, seed=n))
        assert a[2] == b[2] and tiled.counters() == rt.counters() and mismatch(a[0], b[0]) == 0 and np.array_equal(a[1], b[1]), n
    tiled.close()
    # the same kernel on small scenes (their records kept out of LDS): a single tile, a partly filled last chunk
    monkeypatch.setenv("RT_AMD_MAX_LDS_SPHERES", "16")
    small = R.Renderer(0, camera_ray_lists=rt.camera_ray_lists)
    for n in (20, 33, 487, 512, 513):
        table = R.final_scene_table()[:n] if n <= 487 else _random_scene(rng, n)
        for r in (small, rt):
            r.upload_scene(table)
            r.set_camera(R.final_camera(72, 40))
        a, b = small.render(small.params(72, 40, 3, seed=n)), rt.render(rt.params(72, 40, 3, seed=n))
        assert a[2] == b[2] and mismatch(a[0], b[0]) == 0 and np.array_equal(a[1], b[1]), n
    small.close()


def test_filter_variants_agree(rt, monkeypatch):
    """The plane pre-filter on the matrix pipe (default), the same filter on the vector ALU and the full-discriminant
    filter are all conservative: same image, same ray count as the literal kernel.  Near the horizon the reference's own
    discriminant rounding lets it hit spheres whose centre is up to 1e-4 farther than r from the ray; the slack of every
    filter covers it."""
    imgs = {}
    for filt in ("mfma", "plane", "disc"):
        monkeypatch.setenv("RT_AMD_FILTER", filt)
        r2 = R.Renderer(0, camera_ray_lists=rt.camera_ray_lists)
        r2.upload_scene(R.final_scene_table())
        r2.set_camera(R.final_camera(1200, 800))
        # rows 240..287: the horizon band where grazing hits occur
        p = r2.params(1200, 800, 24, seed=0, band_rows=48, band_first=5, band_step=1000)
        imgs[filt] = r2.render(p)[:3]
        if filt == "plane":
            imgs["literal"] = r2.render(r2.params(1200, 800, 24, seed=0, band_rows=48, band_first=5, band_step=1000, mode=R.MODE_PIXEL))[:3]
        r2.close()
    for k in ("mfma", "plane", "disc"):
        assert imgs[k][2] == imgs["literal"][2]
        assert np.array_equal(imgs[k][0].view(np.uint32), imgs["literal"][0].view(np.uint32))


def test_edge_cases(rt):
    rt.upload_scene(R.three_sphere_table())
    rt.set_camera(R.three_camera(7, 5))
    # ragged sizes (not multiples of the 8x8 wave tile), one sample, depth 0 and 1
    scene = O.SceneArrays.from_table(O.three_sphere_table())
    ocam = O.three_camera(7, 5)
    rgb, irgb, rays, _ = rt.render(rt.params(7, 5, 1, seed=5))
    want, wq, wrays = O.render(scene, ocam, 7, 5, 1, O.usrc_philox(5))
    assert rays == wrays and same(rgb, want) and np.array_equal(irgb, wq)
    d0, _, r0, _ = rt.render(rt.params(7, 5, 2, seed=5, max_depth=0))
    assert r0 == 7 * 5 * 2           # depth cap 0: one query per sample, hits are black
    # a band selection that owns nothing
    p = rt.params(7, 5, 1, seed=5, band_rows=8, band_first=3, band_step=4)
    assert rt.owned_rows(p) == 0
    e, _, re, _ = rt.render(p)
    assert e.shape[0] == 0 and re == 0
    # single sphere, single pixel
    rt.upload_scene(np.array([[0, 0, -1, 0.5, 0, 0.5, 0.5, 0.5, 0]], dtype=np.float64))
    rt.set_camera(R.three_camera(1, 1))
    one, _, r1, _ = rt.render(rt.params(1, 1, 4, seed=0))
    assert one.shape == (1, 1, 3) and r1 >= 4


def test_depth_cap_and_second_context(rt):
    """max_depth other than the reference's 50 against the oracle with the same cap; two contexts side by side."""
    scene = O.SceneArrays.from_table(O.three_sphere_table(True))
    ocam = O.three_camera(48, 27)
    rt.upload_scene(R.three_sphere_table(True))
    rt.set_camera(R.three_camera(48, 27))
    other = R.Renderer(0)
    other.upload_scene(R.final_scene_table())
    other.set_camera(R.final_camera(48, 27))
    try:
        for cap in (1, 3, 7):
            O.lib().orc_set_max_depth(cap)
            want, wq, wrays = O.render(scene, ocam, 48, 27, 6, O.usrc_philox(21))
            for mode in (R.MODE_MEGAKERNEL, R.MODE_WAVEFRONT, R.MODE_PIXEL, R.MODE_BVH):
                rgb, irgb, rays, _ = rt.render(rt.params(48, 27, 6, seed=21, max_depth=cap, mode=mode))
                assert rays == wrays and same(rgb, want) and np.array_equal(irgb, wq), (cap, mode)
    finally:
        O.lib().orc_set_max_depth(50)
    # the second context renders its own scene unaffected by the first
    a, _, ra, _ = other.render(other.params(48, 27, 2, seed=1))
    rt.upload_scene(R.final_scene_table())
    rt.set_camera(R.final_camera(48, 27))
    b, _, rb, _ = rt.render(rt.params(48, 27, 2, seed=1))
    assert ra == rb and same(a, b)
    other.close()


def test_error_behaviour(rt):
    fresh = R.Renderer(0)
    with pytest.raises(R.RtError):
        fresh.render(fresh.params(8, 8, 1))                       # no scene / camera yet
    with pytest.raises(R.RtError):
        fresh.upload_scene(np.array([[0, 0, 0, 1, 7, 1, 1, 1, 0]], dtype=np.float64))   # unknown material
    fresh.upload_scene(R.three_sphere_table())
    fresh.set_camera(R.three_camera(8, 8))
    with pytest.raises(R.RtError):
        fresh.render(fresh.params(0, 8, 1))
    with pytest.raises(R.RtError):
        fresh.render(fresh.params(8, 8, 1, max_depth=65))
    fresh.close()
    with pytest.raises(R.RtError):
        R.Renderer(99)
    # the entry points added around the render call report misuse the same way
    fresh = R.Renderer(0)
    with pytest.raises(R.RtError) as e:
        fresh.counters()                                          # nothing rendered yet
    assert e.value.code == -4
    with pytest.raises(R.RtError):
        fresh.encode_ppm_device(0, 8, 8)                          # null image
    fresh.upload_scene(R.three_sphere_table())
    fresh.set_camera(R.three_camera(8, 8))
    with pytest.raises(R.RtError):
        fresh.render(fresh.params(8, 8, 1, mode=7))               # unknown kernel organisation
    fresh.close()


def test_ppm_text_encoded_on_the_gpu_equals_the_host_formatter(rt, tmp_path):
    """output.ppm (main.cpp:95-97,143-144) produced from the device-resident image: same bytes as the host formatter."""
    import torch
    rng = np.random.default_rng(11)
    cases = []
    # every value 0..255 and the values a NaN or an unclamped radiance can produce
    special = np.array([0, 9, 10, 99, 100, 255, 256, 999, 1000, 65535, -1, -10, -2147483648, 2147483647, 1000000000, 999999999], dtype=np.int32)
    for (w, h) in [(1, 1), (3, 1), (7, 5), (341, 3), (1024, 1), (1025, 1), (257, 129), (1200, 800)]:
        img = rng.integers(0, 256, size=(h, w, 3), dtype=np.int32)
        if w * h >= 16:
            idx = rng.integers(0, w * h * 3, size=min(64, w * h))
            img.reshape(-1)[idx] = special[rng.integers(0, len(special), size=len(idx))]
        cases.append(img)
    cases.append(np.full((33, 65, 3), -2147483648, dtype=np.int32))      # longest possible lines
    cases.append(np.zeros((64, 64, 3), dtype=np.int32))                  # shortest
    for img in cases:
        h, w = img.shape[0], img.shape[1]
        want = R.format_ppm(img)
        d_img = torch.from_numpy(img).cuda()
        need = rt.encode_ppm_device(d_img.data_ptr(), w, h)
        assert need == len(want)
        # a text buffer at an odd address, guarded on both sides
        d_text = torch.full((need + 64,), 0x7e, dtype=torch.uint8, device="cuda")
        for lead in (0, 1, 2, 3):
            d_text.fill_(0x7e)
            got_len = rt.encode_ppm_device(d_img.data_ptr(), w, h, d_text.data_ptr() + 16 + lead, need)
            torch.cuda.synchronize()
            host = d_text.cpu().numpy()
            assert got_len == need
            assert host[16 + lead:16 + lead + need].tobytes() == want
            assert (host[:16 + lead] == 0x7e).all() and (host[16 + lead + need:] == 0x7e).all()
        # capacity one byte short: nothing written, the need reported
        d_text.fill_(0x7e)
        with pytest.raises(R.RtError) as e:
            rt.encode_ppm_device(d_img.data_ptr(), w, h, d_text.data_ptr(), need - 1)
        assert e.value.code == -5
        torch.cuda.synchronize()
        assert (d_text.cpu().numpy() == 0x7e).all()
    # file form, from a rendered image
    rt.upload_scene(R.final_scene_table())
    rt.set_camera(R.final_camera(96, 64))
    p = rt.params(96, 64, 2, seed=4)
    d_irgb = torch.zeros((64, 96, 3), dtype=torch.int32, device="cuda")
    rt.render_device(p, 0, d_irgb.data_ptr())
    rt.stats()
    path = str(tmp_path / "gpu.ppm")
    rt.write_ppm_device(path, d_irgb.data_ptr(), 96, 64)
    assert open(path, "rb").read() == R.format_ppm(d_irgb.cpu().numpy())
    with pytest.raises(R.RtError):
        rt.write_ppm_device("/nonexistent_dir/x.ppm", d_irgb.data_ptr(), 96, 64)


def _random_scene(rng, n):
    """Spheres with the awkward cases the reference's code accepts without complaint: negative radii (hollow
    glass, main.cpp:107), exact duplicates (ties go to the lower index, hittablelist.cpp:21-29), overlapping and
    nested spheres, a camera that may sit inside one, fuzz above the constructor's clamp, odd refraction indices."""
    t = np.zeros((n, 9), dtype=np.float64)
    t[:, 0:3] = rng.uniform(-3.0, 3.0, size=(n, 3))
    t[:, 3] = np.exp(rng.uniform(np.log(0.02), np.log(2.5), size=n))
    t[:, 4] = rng.integers(0, 3, size=n)
    t[:, 5:8] = rng.uniform(0.05, 1.0, size=(n, 3))
    for k in range(n):
        if t[k, 4] == R.METAL:
            t[k, 8] = min(rng.choice([0.0, 0.3, 1.0, 1.7]), 1.0)        # Metal's constructor clamps fuzz to 1 (material.h:73-83)
        elif t[k, 4] == R.DIELECTRIC:
            t[k, 8] = rng.choice([1.5, 0.8, 1.0, 2.4])
            t[k, 5:8] = 1.0
            if rng.random() < 0.4:
                t[k, 3] = -t[k, 3]
        if k > 0 and rng.random() < 0.15:                               # exact duplicate of an earlier sphere, other material
            src = rng.integers(0, k)
            t[k, 0:4] = t[src, 0:4]
    if n > 3 and rng.random() < 0.5:
        t[0] = [0.0, -1000.0, 0.0, 1000.0, R.LAMBERTIAN, 0.5, 0.5, 0.5, 0.0]
    return np.asarray(t, dtype=np.float32).astype(np.float64)


def test_random_scenes_match_oracle_in_all_modes(rt):
    """Seeded random scenes and cameras, small images: the three kernel organisations against the CPU oracle,
    bit for bit (image, quantised integers, ray count and the RenderStats counters)."""
    rng = np.random.default_rng(20240607)
    W, H, S = 40, 24, 3
    for case, n in enumerate([1, 2, 5, 31, 32, 33, 64, 65, 200, 513, 1000, 1536, 1537, 1600, 4000, 4097]):      # 1000 / 1536: more than 64 KB of LDS per block; 4000 / 4097: either side of the wavefront kernel's LDS limit
        table = _random_scene(rng, n)
        look_from = rng.uniform(-4.0, 4.0, size=3)
        look_at = rng.uniform(-1.0, 1.0, size=3)
        vfov = float(rng.choice([20.0, 60.0, 110.0]))
        aperture = float(rng.choice([0.0, 0.3]))
        ocam = O.make_camera(look_from, look_at, (0.0, 1.0, 0.0), vfov, np.float32(W) / np.float32(H), aperture, float(rng.uniform(1.0, 6.0)))
        pod = R.CameraPod()
        for dst, src in (("origin", "origin"), ("lower_left_corner", "llc"), ("horizontal", "horiz"), ("vertical", "vert"), ("u", "u"), ("v", "v"), ("w", "w")):
            vec = getattr(ocam, src)
            getattr(pod, dst)[:] = [vec.x, vec.y, vec.z]
        pod.lens_radius = ocam.lens_radius
        rt.upload_scene(table)
        rt.set_camera(pod)
        seed = 1000 + case
        want, wq, wrays, wcount = O.render_counted(O.SceneArrays.from_table(table), ocam, W, H, S, O.usrc_philox(seed))
        for mode in (R.MODE_MEGAKERNEL, R.MODE_WAVEFRONT, R.MODE_PIXEL, R.MODE_BVH):
            rgb, irgb, rays, _ = rt.render(rt.params(W, H, S, seed=seed, mode=mode))
            assert rays == wrays, (case, n, mode)
            assert rt.counters() == wcount, (case, n, mode)
            assert mismatch(rgb, want) == 0, (case, n, mode, mismatch(rgb, want))
            assert np.array_equal(irgb, wq), (case, n, mode)


def test_last_paths_of_a_launch_are_shared_by_the_block(rt):
    """The end of a launch: once the work counter is dry the waves of a block pool the paths they still hold and deal them out
    afresh before every bounce (a path then runs on other threads than the one it started on, with that thread's hit-index
    stack).  A closed mirror shell (negative radius: the normal points inwards, Metal::Scatter never absorbs, material.h:73-91)
    around camera and spheres keeps every path alive to the depth cap, so launches of a few paths spend nearly all their
    bounces in that phase; images, integers, ray count and counters against the oracle, and larger ones against the
    one-lane-per-pixel kernel."""
    rng = np.random.default_rng(77)
    n = 40
    t = np.zeros((n, 9), dtype=np.float64)
    t[0] = [0.0, 0.0, 0.0, -30.0, R.METAL, 0.97, 0.95, 0.9, 0.0]
    t[1:, 0:3] = rng.uniform(-6.0, 6.0, size=(n - 1, 3))
    t[1:, 3] = rng.uniform(0.3, 1.2, size=n - 1)
    t[1:, 4] = rng.integers(0, 3, size=n - 1)
    t[1:, 5:8] = rng.uniform(0.3, 1.0, size=(n - 1, 3))
    for k in range(1, n):
        if t[k, 4] == R.METAL: t[k, 8] = rng.choice([0.0, 0.4])
        elif t[k, 4] == R.DIELECTRIC: t[k, 8] = 1.5; t[k, 5:8] = 1.0
    table = np.asarray(t, dtype=np.float32).astype(np.float64)
    rt.upload_scene(table)
    scene = O.SceneArrays.from_table(table)
    for W, H, S, cap in ((6, 6, 1, 50), (9, 5, 1, 50), (1, 1, 3, 50), (40, 3, 2, 50), (64, 10, 1, 17), (33, 17, 2, 50)):
        ocam = O.make_camera((9.0, 2.0, 7.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 50.0, np.float32(W) / np.float32(H), 0.2, 9.0)
        pod = R.CameraPod()
        for dst, src in (("origin", "origin"), ("lower_left_corner", "llc"), ("horizontal", "horiz"), ("vertical", "vert"), ("u", "u"), ("v", "v"), ("w", "w")):
            vec = getattr(ocam, src)
            getattr(pod, dst)[:] = [vec.x, vec.y, vec.z]
        pod.lens_radius = ocam.lens_radius
        rt.set_camera(pod)
        try:
            O.lib().orc_set_max_depth(cap)
            want, wq, wrays, wcount = O.render_counted(scene, ocam, W, H, S, O.usrc_philox(300 + W))
        finally:
            O.lib().orc_set_max_depth(50)
        assert wrays > 0.8 * W * H * S * (cap + 1), "the shell is meant to keep (nearly) every path alive to the depth cap"
        rgb, irgb, rays, _ = rt.render(rt.params(W, H, S, seed=300 + W, max_depth=cap))
        assert rays == wrays and rt.counters() == wcount, (W, H, S, cap)
        assert mismatch(rgb, want) == 0 and np.array_equal(irgb, wq), (W, H, S, cap)
    # more paths than one block holds, several blocks, ragged edges: against the per-pixel kernel
    for W, H, S in ((96, 64, 3), (250, 130, 1)):
        ocam = O.make_camera((9.0, 2.0, 7.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 50.0, np.float32(W) / np.float32(H), 0.2, 9.0)
        pod = R.CameraPod()
        for dst, src in (("origin", "origin"), ("lower_left_corner", "llc"), ("horizontal", "horiz"), ("vertical", "vert"), ("u", "u"), ("v", "v"), ("w", "w")):
            vec = getattr(ocam, src)
            getattr(pod, dst)[:] = [vec.x, vec.y, vec.z]
        pod.lens_radius = ocam.lens_radius
        rt.set_camera(pod)
        a, ai, ra, _ = rt.render(rt.params(W, H, S, seed=5, mode=R.MODE_MEGAKERNEL))
        ca = rt.counters()
        b, bi, rb, _ = rt.render(rt.params(W, H, S, seed=5, mode=R.MODE_PIXEL))
        assert ra == rb and ca == rt.counters() and mismatch(a, b) == 0 and np.array_equal(ai, bi), (W, H, S)


def test_camera_rays_through_tile_lists_equal_the_full_sweep(rt, monkeypatch):
    """By default the first level of Color() (main.cpp:13-39) for a camera ray is worked out against the list of spheres that the
    rays of its 8x8 tile can reach; RT_AMD_PRIMARY=sweep (read when a context is created) sweeps camera rays like every other
    ray.  The list has to contain every sphere the reference's test could accept, so both contexts, and the per-pixel kernel
    that knows neither lists nor filters, must agree bit for bit where the list is hardest to get right: wide lenses focused
    close, a camera inside spheres (hollow ones too), roots behind the origin (negative tmin), a skewed camera record, rows
    dealt out one by one over three ranks (a tile then spans rows far apart), more candidates than a list holds, and a camera
    that is not finite (no list; such rays hit nothing)."""
    swept = R.Renderer(0, camera_ray_lists=False)
    before = rt.camera_ray_lists
    rt.set_camera_ray_lists(True)          # (one of the fixture's two contexts has them off: this test is about the lists)
    rng = np.random.default_rng(77)
    W, H, S = 104, 60, 2

    def pod_of(look_from, look_at, vfov, aperture, focus):
        ocam = O.make_camera(look_from, look_at, (0.0, 1.0, 0.0), vfov, np.float32(W) / np.float32(H), aperture, focus)
        pod = R.CameraPod()
        for dst, src in (("origin", "origin"), ("lower_left_corner", "llc"), ("horizontal", "horiz"), ("vertical", "vert"), ("u", "u"), ("v", "v"), ("w", "w")):
            vec = getattr(ocam, src)
            getattr(pod, dst)[:] = [vec.x, vec.y, vec.z]
        pod.lens_radius = ocam.lens_radius
        return pod

    cases = []
    for k in range(6):       # wide lenses, close focus, spheres down to the lens
        cases.append((_random_scene(rng, int(rng.choice([40, 300]))), pod_of(rng.uniform(-3, 3, 3), rng.uniform(-1, 1, 3), float(rng.choice([25.0, 70.0])), float(rng.choice([0.5, 2.0, 4.0])), float(rng.uniform(0.4, 5.0))), {}))
    inside = _random_scene(rng, 60)
    inside[0] = [0.2, 0.1, -0.3, 2.5, R.DIELECTRIC, 1.0, 1.0, 1.0, 1.5]           # the camera sits inside this one ...
    inside[1] = [0.2, 0.1, -0.3, -2.3, R.DIELECTRIC, 1.0, 1.0, 1.0, 1.5]          # ... and inside its hollow twin
    cases.append((inside, pod_of((0.0, 0.0, 0.0), (1.0, 0.2, -1.0), 60.0, 0.2, 2.0), {}))
    cases.append((_random_scene(rng, 120), pod_of((2.0, 1.0, 2.0), (0.0, 0.0, 0.0), 50.0, 0.3, 3.0), {"tmin": -0.75}))      # roots behind the origin count
    cases.append((_random_scene(rng, 120), pod_of((2.0, 1.0, 2.0), (0.0, 0.0, 0.0), 50.0, 0.0, 3.0), {"tmin": -1e30}))
    skew = pod_of((3.0, 1.5, 2.0), (0.0, 0.3, 0.0), 40.0, 0.6, 3.5)
    skew.u[:] = [1.7 * skew.u[0] + 0.4 * skew.v[0], 1.7 * skew.u[1] + 0.4 * skew.v[1], 1.7 * skew.u[2] + 0.4 * skew.v[2]]          # neither unit nor orthogonal
    skew.vertical[:] = [skew.vertical[0] + 0.3, skew.vertical[1], skew.vertical[2] - 0.2]
    cases.append((_random_scene(rng, 200), skew, {}))
    cases.append((_random_scene(rng, 200), pod_of((3.0, 1.5, 2.0), (0.0, 0.3, 0.0), 40.0, 0.3, 3.5), {"band_rows": 1, "band_first": 1, "band_step": 3}))
    crowd = _random_scene(rng, 3000)
    crowd[:, 0:3] = rng.uniform(-0.6, 0.6, size=(3000, 3)); crowd[:, 3] = 0.05                                                      # hundreds of spheres behind every tile
    cases.append((np.asarray(crowd, dtype=np.float32).astype(np.float64), pod_of((0.0, 0.0, 4.0), (0.0, 0.0, 0.0), 20.0, 0.1, 4.0), {}))
    broken = pod_of((3.0, 1.5, 2.0), (0.0, 0.3, 0.0), 40.0, 0.3, 3.5)
    broken.horizontal[1] = float("nan")
    cases.append((_random_scene(rng, 50), broken, {}))
    far = pod_of((3.0e4, 1.5e4, 2.0e4), (0.0, 0.0, 0.0), 0.02, 0.0, 4.0e4)          # the float ray's direction is coarse here
    cases.append((_random_scene(rng, 80), far, {}))

    listed = []
    for case, (table, pod, extra) in enumerate(cases):
        results = []
        for r in (rt, swept):
            r.upload_scene(table)
            r.set_camera(pod)
            p = r.params(W, H, S, seed=300 + case, **{k: v for k, v in extra.items() if k != "tmin"})
            if "tmin" in extra:
                p.tmin = extra["tmin"]
            results.append(r.render(p)[:3] + (r.counters(),))
            if r is rt:
                counts, lists = rt.test_tile_lists(p)
                listed.append(float(np.mean(counts >= 0)))
                for c, row in zip(counts, lists):
                    assert c < 0 or (np.all(np.diff(row[:c].astype(np.int64)) > 0) and (c == 0 or row[c - 1] < len(table))), case       # ascending, in range
                p.mode = R.MODE_PIXEL
                results.append(r.render(p)[:3] + (r.counters(),))
        base = results[0]
        for other in results[1:]:
            assert other[2] == base[2], case
            assert other[3] == base[3], case
            assert np.array_equal(other[0].view(np.uint32), base[0].view(np.uint32)), case
            assert np.array_equal(other[1], base[1]), case
    swept.close()
    rt.set_camera_ray_lists(before)
    # the cases are about lists, so most of their tiles must have one: all but the crowd, the camera that is not finite and the far one
    shares = [round(f, 2) for f in listed]
    assert sum(f > 0.9 for f in listed[:11]) >= 9 and min(listed[:11]) > 0.1, shares       # (a four-unit lens over 300 spheres overflows most lists)
    assert listed[11] < 0.8 and listed[12] == 0.0, shares


def test_tile_lists_hold_every_sphere_an_extreme_ray_of_the_tile_hits(rt):
    """The list of a tile against Sphere::Hit itself (sphere.cpp:15-52, restated here in numpy float32, which rounds every
    operation like the reference's unfused code): rays at the corners of the tile's pixel rectangle (jitter 0 and 1 - 2^-24,
    main.cpp:126-127) from points on the rim of the lens and its centre, produced by the library's Camera::GetRay
    (rt_test_get_ray, camera.cpp:46-53), against every sphere.  Every sphere that reports a hit (discriminant > 0 and a root above
    tmin) must be on the tile's list -- all of them, not only the closest one, which is what the image tests see."""
    W, H = 200, 120
    table = R.final_scene_table()
    geom = np.asarray(table, dtype=np.float64)[:, 0:4].astype(np.float32)
    top = np.float32(1.0) - np.float32(2.0 ** -24)
    rim = []
    for k in range(16):
        a = 2.0 * np.pi * k / 16
        dx, dy = np.float32(np.cos(a)), np.float32(np.sin(a))
        while (dx * dx) + (dy * dy) + np.float32(0.0) >= np.float32(1.0):          # RandomUnitInDisk's acceptance test (camera.cpp:31-44)
            dx, dy = dx * np.float32(0.9999999), dy * np.float32(0.9999999)
        rim.append((dx, dy))
    rim.append((np.float32(0.0), np.float32(0.0)))
    rim = np.array(rim, dtype=np.float32)
    for aperture, focus, tmin in ((0.1, 10.0, 0.001), (2.0, 6.0, 0.001), (0.6, 14.0, -50.0)):
        ocam = O.make_camera((13.0, 2.0, 3.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 20.0, np.float32(W) / np.float32(H), aperture, focus)
        pod = R.CameraPod()
        for dst, src in (("origin", "origin"), ("lower_left_corner", "llc"), ("horizontal", "horiz"), ("vertical", "vert"), ("u", "u"), ("v", "v"), ("w", "w")):
            vec = getattr(ocam, src)
            getattr(pod, dst)[:] = [vec.x, vec.y, vec.z]
        pod.lens_radius = ocam.lens_radius
        rt.upload_scene(table)
        rt.set_camera(pod)
        p = rt.params(W, H, 1, tmin=tmin)
        counts, lists = rt.test_tile_lists(p)
        tiles_x = (W + 7) // 8
        assert (counts >= 0).mean() > 0.5, (aperture, (counts >= 0).mean())
        # corner rays of every tile: (s, t) as the render loop forms them, times the lens points
        ss, tt, dd, owner = [], [], [], []
        for tile in range(len(counts)):
            if counts[tile] < 0:
                continue
            ty, tx = divmod(tile, tiles_x)
            i_lo, i_hi = tx * 8, min(tx * 8 + 7, W - 1)
            y_lo, y_hi = ty * 8, min(ty * 8 + 7, H - 1)
            for i, ui in ((i_lo, np.float32(0.0)), (i_hi, top)):
                for y, uj in ((y_lo, top), (y_hi, np.float32(0.0))):
                    j = H - 1 - y
                    fu = np.float32(np.float64(i) + np.float64(ui)) / np.float32(W)
                    fv = np.float32(np.float64(j) + np.float64(uj)) / np.float32(H)
                    ss += [fu] * len(rim); tt += [fv] * len(rim); dd += list(rim); owner += [tile] * len(rim)
        ro, rd = rt.test_get_ray(pod, np.array(ss, np.float32), np.array(tt, np.float32), np.array(dd, np.float32))
        owner = np.array(owner)
        on_list = np.zeros((len(counts), len(geom)), dtype=bool)
        for tile in range(len(counts)):
            on_list[tile, lists[tile, :max(counts[tile], 0)]] = True
        f = np.float32
        hits_seen = 0
        for lo in range(0, len(owner), 4096):
            o, d, own = ro[lo:lo + 4096, None, :], rd[lo:lo + 4096, None, :], owner[lo:lo + 4096]
            oc = o - geom[None, :, 0:3]
            dot = lambda a, b: (a[..., 0] * b[..., 0] + a[..., 1] * b[..., 1]) + a[..., 2] * b[..., 2]
            a_ = dot(d, d); b_ = dot(oc, d); c_ = dot(oc, oc) - geom[None, :, 3] * geom[None, :, 3]
            desc = b_ * b_ - a_ * c_
            with np.errstate(invalid="ignore", divide="ignore"):
                sq = np.sqrt(np.where(desc > 0, desc, f(0.0)))
                hit = (desc > 0) & ((((-b_ - sq) / a_) > f(tmin)) | (((-b_ + sq) / a_) > f(tmin)))
            assert desc.dtype == np.float32
            missing = hit & ~on_list[own]
            assert not missing.any(), (aperture, np.argwhere(missing)[:5], own[np.argwhere(missing)[:5, 0]])
            hits_seen += int(hit.sum())
        # the check has teeth: the rays do hit spheres (most of them the ground, many a small sphere too), and the lists are short
        assert len(owner) == 68 * int((counts >= 0).sum()) and hits_seen > len(owner) // 2, (len(owner), hits_seen)
        assert on_list.sum() < 0.05 * on_list.size
        print(f"aperture {aperture}: {len(owner)} rays, {hits_seen} ray-sphere hits, all on the lists; mean list {on_list.sum() / max(1, (counts >= 0).sum()):.1f}")


def test_tile_lists_of_the_final_scene_are_short(rt):
    """The benchmark's scene and camera (main.cpp:41-83, 111-116) at 1200x800: every tile has a list, and a list holds a few
    spheres of the 487 (2.1 on average, 11 at most when this was written)."""
    rt.upload_scene(R.final_scene_table())
    rt.set_camera(R.final_camera(1200, 800))
    counts, lists = rt.test_tile_lists(rt.params(1200, 800, 1))
    assert counts.shape == (150 * 100,) and counts.min() >= 0
    assert counts.mean() < 4.0 and counts.max() <= 20, (counts.mean(), counts.max())


def test_full_size_rows_match_the_reference_build(rt):
    """Rows of the 1200x800 image (the size the benchmark is quoted on) rendered by the reference's own code -- the pinned build
    oracle/_ref/ref_harness, compiled from the reference's sources where they lie -- on the keyed stream, against every kernel
    organisation, bit for bit.  The GPU box gets that build with the tree (it is made where the reference's sources are);
    a tree without it fails here rather than passing with the comparison left out (RT_AMD_ALLOW_MISSING_REF=1 skips instead)."""
    ref_band = _require_ref_build()
    W, H, S, seed = 1200, 800, 3, 5
    rt.upload_scene(R.final_scene_table())
    rt.set_camera(R.final_camera(W, H))
    bands = [(j0, 4) + ref_band.keyed_band_exact(W, H, S, j0, 4, seed) for j0 in (398, 520, 120)]      # spheres, horizon, foreground
    for mode in (R.MODE_MEGAKERNEL, R.MODE_BVH, R.MODE_WAVEFRONT, R.MODE_PIXEL):
        rgb, _, _, _ = rt.render(rt.params(W, H, S, seed=seed, mode=mode))
        for j0, rows, want, _ in bands:
            assert mismatch(rgb[H - (j0 + rows):H - j0], want) == 0, (mode, j0)


def _require_ref_build():
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import ref_band
    if not ref_band.exact_available():
        if os.environ.get("RT_AMD_ALLOW_MISSING_REF") == "1":
            pytest.skip("oracle/_ref/ref_harness not built")
        pytest.fail("oracle/_ref/ref_harness (the pinned build of the reference, made by `make -C oracle` where /root/reference exists) "
                    "is not in this tree: the comparison with the reference's own code cannot run")
    return ref_band


def test_equal_radii_scenes_match_oracle_in_all_modes(rt):
    """Scenes whose spheres (nearly) all have the same radius: the plane filter stores whole chunks of 32 such spheres in a
    normalised form with one threshold for the chunk.  Equal, nearly equal (within 0.1 % and within 10 %) and equal with a
    few odd ones mixed in; every kernel organisation against the CPU oracle, bit for bit."""
    rng = np.random.default_rng(4242)
    W, H, S = 40, 24, 3
    for case, (n, spread_r) in enumerate([(64, 0.0), (200, 0.001), (513, 0.1), (96, 0.0)]):
        table = _random_scene(rng, n)
        r0 = float(rng.choice([0.05, 0.2, 0.6]))
        sign = np.sign(table[:, 3])
        table[:, 3] = sign * r0 * (1.0 - spread_r * rng.random(n))
        if case == 3:
            table[rng.choice(n, size=5, replace=False), 3] = [1.5, 0.01, 3.0, 0.0, 0.2]      # chunks with odd ones stay in the plain form
        table = np.asarray(table, dtype=np.float32).astype(np.float64)
        look_from = rng.uniform(-5.0, 5.0, size=3)
        ocam = O.make_camera(look_from, rng.uniform(-1.0, 1.0, size=3), (0.0, 1.0, 0.0), float(rng.choice([30.0, 80.0])), np.float32(W) / np.float32(H), float(rng.choice([0.0, 0.2])), 4.0)
        pod = R.CameraPod()
        for dst, src in (("origin", "origin"), ("lower_left_corner", "llc"), ("horizontal", "horiz"), ("vertical", "vert"), ("u", "u"), ("v", "v"), ("w", "w")):
            vec = getattr(ocam, src)
            getattr(pod, dst)[:] = [vec.x, vec.y, vec.z]
        pod.lens_radius = ocam.lens_radius
        rt.upload_scene(table)
        rt.set_camera(pod)
        seed = 6000 + case
        want, wq, wrays, wcount = O.render_counted(O.SceneArrays.from_table(table), ocam, W, H, S, O.usrc_philox(seed))
        for mode in (R.MODE_MEGAKERNEL, R.MODE_WAVEFRONT, R.MODE_PIXEL, R.MODE_BVH):
            rgb, irgb, rays, _ = rt.render(rt.params(W, H, S, seed=seed, mode=mode))
            assert rays == wrays, (case, n, mode)
            assert mismatch(rgb, want) == 0, (case, n, mode, mismatch(rgb, want))
            assert np.array_equal(irgb, wq), (case, n, mode)


def test_extreme_radii_match_oracle_in_all_modes(rt):
    """Radii far outside the ordinary range: points (r = 0), specks of 1e-9 .. 1e-5, spheres of 6e5 and 1e7 units (the
    kernels' plane filter does not judge such spheres itself and must pass them on), mixed with ordinary ones, seen from
    near and from far away.  Every kernel organisation against the CPU oracle, bit for bit."""
    rng = np.random.default_rng(77)
    W, H, S = 40, 24, 3
    for case, n in enumerate([6, 40, 70, 300]):
        table = _random_scene(rng, n)
        odd = rng.choice(n, size=max(3, n // 4), replace=False)
        table[odd, 3] = rng.choice([0.0, 1e-9, 1e-7, 1e-5, -1e-6, 6e5, 1e7], size=len(odd))
        big = np.abs(table[:, 3]) > 1e3
        table[big, 1] = -np.abs(table[big, 3]) - rng.uniform(0.0, 2.0, size=big.sum())      # huge ones below the scene, like a ground
        table = np.asarray(table, dtype=np.float32).astype(np.float64)
        dist = float(rng.choice([3.0, 8.0, 500.0]))
        direction = rng.normal(size=3); direction[1] = abs(direction[1]) + 0.2; direction /= np.linalg.norm(direction)
        ocam = O.make_camera(direction * dist, (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 40.0 if dist < 100 else 1.5, np.float32(W) / np.float32(H), 0.0, dist)
        pod = R.CameraPod()
        for dst, src in (("origin", "origin"), ("lower_left_corner", "llc"), ("horizontal", "horiz"), ("vertical", "vert"), ("u", "u"), ("v", "v"), ("w", "w")):
            vec = getattr(ocam, src)
            getattr(pod, dst)[:] = [vec.x, vec.y, vec.z]
        pod.lens_radius = ocam.lens_radius
        rt.upload_scene(table)
        rt.set_camera(pod)
        seed = 4000 + case
        want, wq, wrays, wcount = O.render_counted(O.SceneArrays.from_table(table), ocam, W, H, S, O.usrc_philox(seed))
        for mode in (R.MODE_MEGAKERNEL, R.MODE_WAVEFRONT, R.MODE_PIXEL, R.MODE_BVH):
            rgb, irgb, rays, _ = rt.render(rt.params(W, H, S, seed=seed, mode=mode))
            assert rays == wrays, (case, n, mode)
            assert mismatch(rgb, want) == 0, (case, n, mode, mismatch(rgb, want))
            assert np.array_equal(irgb, wq), (case, n, mode)


def test_tile_work_order_is_invisible_and_follows_scene_camera_and_rows(rt, monkeypatch):
    """The order in which tiles are worked on (longest paths first, kept between calls) never shows in the image:
    same bits as the image-order variant, and a new camera, scene or row selection is picked up by the next call."""
    def fresh(scene, cam, p_kw, env=None):
        if env:
            monkeypatch.setenv("RT_AMD_TILE_ORDER", env)
        r = R.Renderer(0)
        monkeypatch.delenv("RT_AMD_TILE_ORDER", raising=False)
        r.upload_scene(scene)
        r.set_camera(cam)
        out = r.render(r.params(**p_kw))[:3]
        r.close()
        return out

    final, three = R.final_scene_table(), R.three_sphere_table(True)
    cam_a, cam_b = R.final_camera(120, 72), R.Camera((3.0, 1.5, -6.0), (0.0, 0.5, 0.0), (0.0, 1.0, 0.0), 40.0, np.float32(120) / np.float32(72), 0.05, 6.0)
    kw = dict(width=120, height=72, spp=3, seed=12)
    rt.upload_scene(final)
    rt.set_camera(cam_a)
    first = rt.render(rt.params(**kw))[:3]
    again = rt.render(rt.params(**kw))[:3]                       # second call reuses the stored order
    ref = fresh(final, cam_a, kw, env="image")
    for got in (first, again):
        assert got[2] == ref[2] and mismatch(got[0], ref[0]) == 0 and np.array_equal(got[1], ref[1])
    rt.set_camera(cam_b)                                         # new camera, same context
    got = rt.render(rt.params(**kw))[:3]
    ref = fresh(final, cam_b, kw, env="image")
    assert got[2] == ref[2] and mismatch(got[0], ref[0]) == 0
    rt.upload_scene(three)                                       # new scene, same camera
    got = rt.render(rt.params(**kw))[:3]
    ref = fresh(three, cam_b, kw, env="image")
    assert got[2] == ref[2] and mismatch(got[0], ref[0]) == 0
    kw_rows = dict(kw, band_rows=5, band_first=1, band_step=3)   # other rows, same scene and camera
    got = rt.render(rt.params(**kw_rows))[:3]
    ref = fresh(three, cam_b, kw_rows, env="image")
    assert got[0].shape == ref[0].shape and got[2] == ref[2] and mismatch(got[0], ref[0]) == 0


@pytest.mark.parametrize("distance", [60.0, 400.0, 3000.0, 20000.0])
def test_far_telephoto_camera_where_the_reference_discriminant_is_noisy(rt, distance):
    """Seen from far away through a narrow lens, the reference's float discriminant b*b - a*c loses its digits:
    it reports hits for lines that pass well outside a sphere.  Filters and hierarchy boxes carry an allowance for
    exactly that; here the sphere field is looked at from up to 20 000 units, and the filtered megakernel, the
    wavefront kernels and the hierarchy walk must still agree bit for bit with the literal per-sphere kernel."""
    W, H, S = 64, 40, 4
    rt.upload_scene(R.final_scene_table())
    direction = np.array([13.0, 2.0, 3.0]) / np.linalg.norm([13.0, 2.0, 3.0])
    look_from = direction * distance
    vfov = float(np.degrees(2.0 * np.arctan(6.0 / distance)))         # the field of small spheres fills the frame
    cam = R.Camera(look_from, (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), vfov, np.float32(W) / np.float32(H), 0.0, distance)
    rt.set_camera(cam)
    lit = rt.render(rt.params(W, H, S, seed=77, mode=R.MODE_PIXEL))
    assert np.isfinite(lit[0]).all()
    for mode in (R.MODE_MEGAKERNEL, R.MODE_WAVEFRONT, R.MODE_BVH):
        got = rt.render(rt.params(W, H, S, seed=77, mode=mode))
        assert got[2] == lit[2], (distance, mode)
        assert mismatch(got[0], lit[0]) == 0, (distance, mode, mismatch(got[0], lit[0]))


@pytest.mark.parametrize("mode", [R.MODE_MEGAKERNEL, R.MODE_WAVEFRONT, R.MODE_BVH])
def test_sample_batches_accumulate_in_sample_order(monkeypatch, mode):
    """Large jobs render their samples in batches with a running per-pixel sum; the sum must still run over the
    samples in ascending order (main.cpp:124-134).  A small cap on the sample buffer forces many batches here."""
    m = _meta()["philox_final_40x30x16_ap0"]
    want = np.load(os.path.join(G, "img_philox_final_40x30x16_ap0.npy"))
    monkeypatch.setenv("RT_AMD_SAMPLE_BUFFER_KB", "64")          # 40x32 padded pixels x 12 B = 15 KB per sample: 4 samples per batch
    r = R.Renderer(0)
    monkeypatch.delenv("RT_AMD_SAMPLE_BUFFER_KB")
    _setup(r, m)
    rgb, irgb, rays, _ = r.render(r.params(m["nx"], m["ny"], m["ns"], seed=m["seed"], mode=mode))
    assert rays == m["rays"] and r.counters() == (m["rays"], m["hits"], m["scatters"])
    assert mismatch(rgb, want) == 0
    if mode in (R.MODE_MEGAKERNEL, R.MODE_BVH):
        # the same through tile shares, whose running sums are kept per work tile: three shares, several batches each
        import torch
        from raytracing_amd import dist as rdist
        p = r.params(m["nx"], m["ny"], m["ns"], seed=m["seed"], mode=mode)
        tiles, ids, total, counted = [], [], 0, [0, 0, 0]
        for k in range(3):
            cnt = r.owned_tiles(p, k, 3)
            t = torch.zeros((cnt, 64, 3), dtype=torch.float32, device="cuda"); i = torch.zeros(cnt, dtype=torch.int32, device="cuda")
            r.render_tiles_device(p, k, 3, d_rgb_mean_tiles=t.data_ptr(), d_tile_ids=i.data_ptr())
            total += r.stats()[0]
            counted = [a + b for a, b in zip(counted, r.counters())]
            tiles.append(t); ids.append(i)
        img = rdist.assemble_tiles(torch.cat(tiles), torch.cat(ids), m["nx"], m["ny"]).cpu().numpy()
        assert total == m["rays"] and tuple(counted) == (m["rays"], m["hits"], m["scatters"]) and mismatch(img, want) == 0
    # a cap below one sample of the image is an error, not a crash
    monkeypatch.setenv("RT_AMD_SAMPLE_BUFFER_KB", "8")
    tiny = R.Renderer(0)
    monkeypatch.delenv("RT_AMD_SAMPLE_BUFFER_KB")
    _setup(tiny, m)
    with pytest.raises(R.RtError) as e:
        tiny.render(tiny.params(m["nx"], m["ny"], m["ns"], seed=m["seed"], mode=mode))
    assert e.value.code == -5
    r.close(); tiny.close()


@pytest.mark.parametrize("mode", [R.MODE_MEGAKERNEL, R.MODE_BVH])
def test_camera_rays_from_their_own_kernel(monkeypatch, mode):
    """With camera rays swept like every other ray, the camera rays of a sample batch come from a kernel of their own, through a
    queue in device memory (main.cpp:118-133 with Camera::GetRay, camera.cpp:46-53).  Same image, ray count and counters as with
    the rays generated inside the render kernel and as the golden image: on a ragged size (padding pixels leave empty queue rows),
    on interleaved rows, over several batches forced by a small queue cap, and through tile shares."""
    import torch
    from raytracing_amd import dist as rdist
    m = _meta()["philox_final_40x30x16_ap0"]
    want = np.load(os.path.join(G, "img_philox_final_40x30x16_ap0.npy"))
    monkeypatch.setenv("RT_AMD_QUEUE_KB", "256")             # 40x32 padded pixels x 32 B = 40 KB per sample: 6 samples per batch (104x64 further down: one)
    queued = R.Renderer(0, camera_ray_lists=False)
    monkeypatch.delenv("RT_AMD_QUEUE_KB")
    monkeypatch.setenv("RT_AMD_PRIMARY_KERNEL", "0")
    staged = R.Renderer(0, camera_ray_lists=False)
    monkeypatch.delenv("RT_AMD_PRIMARY_KERNEL")
    for r in (queued, staged):
        _setup(r, m)
        rgb, irgb, rays, _ = r.render(r.params(m["nx"], m["ny"], m["ns"], seed=m["seed"], mode=mode))
        assert rays == m["rays"] and r.counters() == (m["rays"], m["hits"], m["scatters"])
        assert mismatch(rgb, want) == 0
    # three tile shares, several batches each
    p = queued.params(m["nx"], m["ny"], m["ns"], seed=m["seed"], mode=mode)
    tiles, ids, total = [], [], 0
    for k in range(3):
        cnt = queued.owned_tiles(p, k, 3)
        t = torch.zeros((cnt, 64, 3), dtype=torch.float32, device="cuda"); i = torch.zeros(cnt, dtype=torch.int32, device="cuda")
        queued.render_tiles_device(p, k, 3, d_rgb_mean_tiles=t.data_ptr(), d_tile_ids=i.data_ptr())
        total += queued.stats()[0]
        tiles.append(t); ids.append(i)
    img = rdist.assemble_tiles(torch.cat(tiles), torch.cat(ids), m["nx"], m["ny"]).cpu().numpy()
    assert total == m["rays"] and mismatch(img, want) == 0
    # ragged sizes, interleaved rows, a wide lens: queue against in-kernel staging against the literal kernel
    for r in (queued, staged):
        r.upload_scene(R.final_scene_table())
        r.set_camera(R.final_camera(101, 59))
    for kw in (dict(), dict(band_rows=3, band_first=1, band_step=2), dict(band_rows=1, band_first=6, band_step=7)):
        a = queued.render(queued.params(101, 59, 7, seed=3, mode=mode, **kw))
        b = staged.render(staged.params(101, 59, 7, seed=3, mode=mode, **kw))
        c = staged.render(staged.params(101, 59, 7, seed=3, mode=R.MODE_PIXEL, **kw))
        assert a[2] == b[2] == c[2] and a[2] > 0
        assert np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32)) and np.array_equal(a[0].view(np.uint32), c[0].view(np.uint32))
        assert np.array_equal(a[1], c[1])
    # a queue cap below one sample of the image is an error, not a crash
    monkeypatch.setenv("RT_AMD_QUEUE_KB", "16")
    tiny = R.Renderer(0, camera_ray_lists=False)
    monkeypatch.delenv("RT_AMD_QUEUE_KB")
    _setup(tiny, m)
    with pytest.raises(R.RtError) as e:
        tiny.render(tiny.params(m["nx"], m["ny"], m["ns"], seed=m["seed"], mode=mode))
    assert e.value.code == -5
    for r in (queued, staged, tiny):
        r.close()


def _ref_band():
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import ref_band
    return ref_band


def test_config4_rows_match_oracle_and_reference_build(rt):
    """3840x2160 (the multi-GPU configuration; camera of main.cpp:111-116 at 16:9): two rows at three heights, 2 samples,
    every kernel organisation, against the CPU oracle and against the reference's own code (pinned build) on the keyed stream."""
    W, H, S, seed = 3840, 2160, 2, 17
    scene = O.SceneArrays.from_table(O.load_scene_table())
    ocam = O.final_camera(W, H)
    rt.upload_scene(R.final_scene_table())
    rt.set_camera(R.final_camera(W, H))
    rb = _require_ref_build()
    for y0 in (300, 1100, 1800):                 # sky, sphere field, foreground (rows counted from the top)
        want, wq, wrays = O.render(scene, ocam, W, H, S, O.usrc_philox(seed), H - (y0 + 2), H - y0)
        ref_img, ref_rays = rb.keyed_band_exact(W, H, S, H - (y0 + 2), 2, seed)
        assert ref_rays == wrays and mismatch(ref_img, want) == 0
        for mode in (R.MODE_MEGAKERNEL, R.MODE_WAVEFRONT, R.MODE_PIXEL, R.MODE_BVH):
            p = rt.params(W, H, S, seed=seed, band_rows=2, band_first=y0 // 2, band_step=100000, mode=mode)
            rgb, irgb, rays, _ = rt.render(p)
            assert rgb.shape == (2, W, 3)
            assert rays == wrays and mismatch(rgb, want) == 0 and np.array_equal(irgb, wq), (y0, mode)


def test_config4_full_render_equals_its_eight_way_partition(rt):
    """3840x2160 at 1024 samples per pixel on one GPU (8.5e9 samples, several sample batches with a running sum) against
    the same image rendered as the eight interleaved one-row-band shares the 8-GPU partition gives its ranks: same ray
    count, same integer image."""
    import torch
    W, H, S, seed = 3840, 2160, 1024, 3
    rt.upload_scene(R.final_scene_table())
    rt.set_camera(R.final_camera(W, H))
    whole = torch.zeros((H, W, 3), dtype=torch.int32, device="cuda")
    rt.render_device(rt.params(W, H, S, seed=seed), 0, whole.data_ptr())
    rays_whole, ms = rt.stats()
    assert 2.4 < rays_whole / (W * H * S) < 3.0
    parts = torch.zeros((H, W, 3), dtype=torch.int32, device="cuda")
    total = 0
    for rank in range(8):
        p = rt.params(W, H, S, seed=seed, band_rows=1, band_first=rank, band_step=8)
        rows = rt.owned_rows(p)
        assert rows == H // 8
        share = torch.zeros((rows, W, 3), dtype=torch.int32, device="cuda")
        rt.render_device(p, 0, share.data_ptr())
        r, _ = rt.stats()
        total += r
        parts[rank::8] = share
    assert total == rays_whole
    assert torch.equal(parts, whole)
    assert int(whole.min()) >= 0 and int(whole.max()) <= 255
    # and as the eight TILE shares of the tile partition (every eighth 8x8 tile of the work order per rank)
    img, total = _assemble_tile_shares(rt, rt.params(W, H, S, seed=seed), 8)
    assert total == rays_whole and torch.equal(img, whole)
    print(f"config 4 on one GPU: {ms:.0f} ms, {rays_whole / ms / 1e3:.0f} Mrays/s, checksum {int(whole.to(torch.int64).sum())}")


def test_ppm_text_of_the_reference_image_on_the_gpu(rt):
    """The integers of the output.ppm that the reference's own main() wrote (main.cpp:91-97, 143-147), encoded on the GPU:
    the text has to be that file, byte for byte."""
    import gzip
    import torch
    with gzip.open(os.path.join(G, "reference_main_output_400x300x5.ppm.gz"), "rb") as f:
        text = f.read()
    img = np.array(text.split()[4:], dtype=np.int32).reshape(300, 400, 3)
    d_img = torch.from_numpy(img).cuda()
    need = rt.encode_ppm_device(d_img.data_ptr(), 400, 300)
    assert need == len(text)
    d_text = torch.zeros(need, dtype=torch.uint8, device="cuda")
    rt.encode_ppm_device(d_img.data_ptr(), 400, 300, d_text.data_ptr(), need)
    torch.cuda.synchronize()
    assert d_text.cpu().numpy().tobytes() == text


def test_device_pointer_form_on_a_side_stream(rt):
    """rt_render_device enqueues on the caller's stream and returns; outputs land in caller-owned device memory."""
    import torch
    rt.upload_scene(R.final_scene_table())
    rt.set_camera(R.final_camera(96, 64))
    p = rt.params(96, 64, 3, seed=31)
    want_rgb, want_i, want_rays, _ = rt.render(p)
    d_rgb = torch.zeros((64, 96, 3), dtype=torch.float32, device="cuda")      # (first: torch initialises its GPU state lazily)
    d_i = torch.zeros((64, 96, 3), dtype=torch.int32, device="cuda")
    side = torch.cuda.Stream()
    for mode in (R.MODE_MEGAKERNEL, R.MODE_WAVEFRONT, R.MODE_PIXEL, R.MODE_BVH):
        d_rgb.zero_(); d_i.zero_()
        torch.cuda.synchronize()
        rt.render_device(rt.params(96, 64, 3, seed=31, mode=mode), d_rgb.data_ptr(), d_i.data_ptr(), side.cuda_stream)
        rays, ms = rt.stats()                                     # waits for the work on the side stream
        assert rays == want_rays and ms > 0.0
        assert mismatch(d_rgb.cpu().numpy(), want_rgb) == 0 and np.array_equal(d_i.cpu().numpy(), want_i)
    # only one of the two outputs requested
    d_i.zero_(); torch.cuda.synchronize()
    rt.render_device(p, 0, d_i.data_ptr(), side.cuda_stream)
    rt.stats()
    assert np.array_equal(d_i.cpu().numpy(), want_i)
    d_rgb.zero_(); torch.cuda.synchronize()
    rt.render_device(p, d_rgb.data_ptr(), 0, side.cuda_stream)
    rt.stats()
    assert mismatch(d_rgb.cpu().numpy(), want_rgb) == 0


def test_two_streams_on_one_context(rt):
    """Two renders enqueued back to back on different streams share the context's buffers (counters, samples, staging rows):
    the second has to wait for the first.  Both images must come out right."""
    import torch
    rt.upload_scene(R.final_scene_table())
    rt.set_camera(R.final_camera(320, 200))
    pa, pb = rt.params(320, 200, 24, seed=5), rt.params(320, 200, 2, seed=6)
    want_a, want_b = rt.render(pa), rt.render(pb)
    a = torch.zeros((200, 320, 3), dtype=torch.float32, device="cuda")
    b = torch.zeros((200, 320, 3), dtype=torch.float32, device="cuda")
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    for _ in range(3):
        a.zero_(); b.zero_(); torch.cuda.synchronize()
        rt.render_device(pa, a.data_ptr(), 0, sa.cuda_stream)        # long one first
        rt.render_device(pb, b.data_ptr(), 0, sb.cuda_stream)        # at once on another stream
        rays_b, _ = rt.stats()
        torch.cuda.synchronize()
        assert rays_b == want_b[2]
        assert mismatch(a.cpu().numpy(), want_a[0]) == 0 and mismatch(b.cpu().numpy(), want_b[0]) == 0
    # the PPM encoder's block offsets are context state too
    ia = torch.from_numpy(want_a[1]).cuda(); ib = torch.from_numpy(want_b[1][:50]).cuda()
    na, nb = rt.encode_ppm_device(ia.data_ptr(), 320, 200), rt.encode_ppm_device(ib.data_ptr(), 320, 50)
    ta, tb = torch.zeros(na, dtype=torch.uint8, device="cuda"), torch.zeros(nb, dtype=torch.uint8, device="cuda")
    rt.encode_ppm_device(ia.data_ptr(), 320, 200, ta.data_ptr(), na, sa.cuda_stream)
    rt.encode_ppm_device(ib.data_ptr(), 320, 50, tb.data_ptr(), nb, sb.cuda_stream)
    torch.cuda.synchronize()
    assert ta.cpu().numpy().tobytes() == R.format_ppm(want_a[1]) and tb.cpu().numpy().tobytes() == R.format_ppm(want_b[1][:50])


def test_non_finite_spheres_are_never_hit(rt):
    """Spheres whose centre or radius is NaN or infinite: no ray hits them in the reference's arithmetic (sphere.cpp:17-50), and
    they must neither disturb the filters nor the hierarchy.  Every kernel organisation against the CPU oracle."""
    rng = np.random.default_rng(99)
    table = _random_scene(rng, 90)
    table[[3, 17, 40], 0] = [np.nan, np.inf, -np.inf]
    table[[5, 50], 1] = [np.inf, np.nan]
    table[[60, 61], 2] = [np.nan, np.inf]
    table[[70, 71, 72], 3] = [np.nan, np.inf, -np.inf]
    W, H, S = 40, 24, 3
    ocam = O.make_camera((3.0, 2.0, 4.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 50.0, np.float32(W) / np.float32(H), 0.1, 5.0)
    pod = R.CameraPod()
    for dst, src in (("origin", "origin"), ("lower_left_corner", "llc"), ("horizontal", "horiz"), ("vertical", "vert"), ("u", "u"), ("v", "v"), ("w", "w")):
        vec = getattr(ocam, src)
        getattr(pod, dst)[:] = [vec.x, vec.y, vec.z]
    pod.lens_radius = ocam.lens_radius
    rt.upload_scene(table)
    rt.set_camera(pod)
    want, wq, wrays, wcount = O.render_counted(O.SceneArrays.from_table(table), ocam, W, H, S, O.usrc_philox(31))
    for mode in (R.MODE_MEGAKERNEL, R.MODE_WAVEFRONT, R.MODE_PIXEL, R.MODE_BVH):
        rgb, irgb, rays, _ = rt.render(rt.params(W, H, S, seed=31, mode=mode))
        assert rays == wrays and rt.counters() == wcount, mode
        assert mismatch(rgb, want) == 0 and np.array_equal(irgb, wq), mode


def test_render_multi_assembles_the_single_context_image(rt):
    """rt_render_multi: several contexts (here on whatever GPUs are visible, and several on GPU 0) render interleaved bands on
    one host thread each and copy their rows into the caller's image; the result is the single-context image bit for bit."""
    table = R.final_scene_table()
    W, H, S = 200, 131, 3                       # 131 rows: a short last band for every band height used below
    rt.upload_scene(table)
    rt.set_camera(R.final_camera(W, H))
    want = rt.render(rt.params(W, H, S, seed=8))
    devices = R.device_count()
    for n, band in ((1, 8), (devices, 1), (3, 1), (4, 7), (5, 40)):
        rs = []
        for k in range(n):
            r = R.Renderer(k % devices)
            r.upload_scene(table)
            r.set_camera(R.final_camera(W, H))
            rs.append(r)
        rgb, irgb, rays, ms = R.render_multi(rs, rt.params(W, H, S, seed=8, band_rows=band))
        assert rays == want[2] and len(ms) == n and all(m > 0 for m in ms[:min(n, (H + band - 1) // band)])
        assert mismatch(rgb, want[0]) == 0 and np.array_equal(irgb, want[1]), (n, band)
        for r in rs:
            r.close()
    # misuse is reported, not crashed on
    fresh = R.Renderer(0)
    with pytest.raises(R.RtError) as e:
        R.render_multi([rt, fresh], rt.params(W, H, S))
    assert e.value.code == -4
    with pytest.raises(R.RtError):
        R.render_multi([rt, rt], rt.params(W, H, S))
    fresh.close()


def test_config2_at_its_stated_size(rt):
    """Config 2 as BASELINE.json states it: the three-sphere material scene of performance_comparison.cpp:45-54 (camera :66-72) at
    400x225 and 100 samples per pixel.  The megakernel against the per-pixel kernel (which knows neither filters nor lists) bit
    for bit with the same ray count and RenderStats counters, and the image against the reference's own 1024-sample render of the
    scene (100x56, libc random stream): each reference pixel covers a 4x4 block of this image, so the block means are compared;
    bound as in test_statistical_image_bound_vs_high_spp_reference with n1 = 1600 samples per block, plus the mean colour."""
    W, H, S = 400, 225, 100
    rt.upload_scene(R.three_sphere_table())
    rt.set_camera(R.three_camera(W, H))
    rgb, irgb, rays, _ = rt.render(rt.params(W, H, S, seed=21))
    counters = rt.counters()
    lit, ilit, rays_lit, _ = rt.render(rt.params(W, H, S, seed=21, mode=R.MODE_PIXEL))
    assert rays == rays_lit and counters == rt.counters()
    assert mismatch(rgb, lit) == 0 and np.array_equal(irgb, ilit)
    assert 1.5 < rays / (W * H * S) < 4.0 and irgb.min() >= 0 and irgb.max() <= 255
    # rows 80..140 from the top (through the three spheres) against the CPU oracle at the full sample count, bit for bit
    scene = O.SceneArrays.from_table(O.three_sphere_table())
    want_rows, wq, _ = O.render(scene, O.three_camera(W, H), W, H, S, O.usrc_philox(21), H - 140, H - 80)
    assert mismatch(rgb[80:140], want_rows) == 0 and np.array_equal(irgb[80:140], wq)
    want = np.load(os.path.join(G, "img_hispp_three_100x56x1024.npy")).astype(np.float64)
    # 225 rows are 56 blocks of 4 rows plus one row (the reference image was rendered at 100x56, i.e. a slightly different aspect
    # and pixel grid): compare the mean colour and the block means of the top-left 400x224 region with a bound widened for the
    # half-pixel shift of the grids
    blocks = rgb[:224].astype(np.float64).reshape(56, 4, 100, 4, 3).mean(axis=(1, 3))
    np.testing.assert_allclose(rgb.reshape(-1, 3).mean(0), want.reshape(-1, 3).mean(0), rtol=0.02)
    mae = np.abs(blocks - want).mean()
    assert mae < 0.02, mae          # (the CPU oracle at 4 samples per pixel gives 0.009 against the same image)


def test_config3_lists_and_sweep_render_the_same_image(rt):
    """Config 3 (1200x800, 500 samples per pixel, 1.27e9 rays): camera rays through the per-tile sphere lists and camera rays swept
    like every other ray give the same integer image and the same ray count (what bench.py reports as image_checksum_equal)."""
    import torch
    W, H, S = 1200, 800, 500
    rt.upload_scene(R.final_scene_table())
    rt.set_camera(R.final_camera(W, H))
    before = rt.camera_ray_lists
    imgs = []
    for lists in (False, True):
        rt.set_camera_ray_lists(lists)
        out = torch.zeros((H, W, 3), dtype=torch.int32, device="cuda")
        rt.render_device(rt.params(W, H, S, seed=0), 0, out.data_ptr())
        rays, _ = rt.stats()
        imgs.append((out, rays, rt.counters()))
    rt.set_camera_ray_lists(before)
    assert imgs[0][1] == imgs[1][1] and imgs[0][2] == imgs[1][2]
    assert torch.equal(imgs[0][0], imgs[1][0])
    assert 2.4 < imgs[0][1] / (W * H * S) < 3.0


def _assemble_tile_shares(rt, p, n):
    """The image of p from its n tile shares (rt_render_tiles_device), assembled as rank 0 of the multi-GPU path does it."""
    import torch
    from raytracing_amd import dist as rdist
    tiles, ids, total = [], [], 0
    for k in range(n):
        cnt = rt.owned_tiles(p, k, n)
        assert cnt == rdist.owned_tile_count(p.width, p.height, k, n)
        t = torch.zeros((max(cnt, 1), 64, 3), dtype=torch.int32, device="cuda")
        i = torch.full((max(cnt, 1),), -1, dtype=torch.int32, device="cuda")
        rt.render_tiles_device(p, k, n, d_irgb_tiles=t.data_ptr(), d_tile_ids=i.data_ptr())
        r, _ = rt.stats()
        total += r
        tiles.append(t[:cnt]); ids.append(i[:cnt])
    return rdist.assemble_tiles(torch.cat(tiles), torch.cat(ids), p.width, p.height), total


def test_tile_shares_assemble_to_the_whole_image(rt):
    """One image dealt out by tiles (the multi-GPU partition): every n-th 8x8 tile of the work order per share.  Shares rendered one
    after the other on this GPU assemble to the single-call image with the same ray count, for widths and heights that are not
    multiples of eight, more shares than some images have tiles, several sample batches, and the hierarchy walk; the share's
    counters follow RenderStats (main_modern.cpp:23-61) with its own pixels as the primary rays."""
    import torch
    for (W, H, S, scene) in ((104, 60, 3, "final"), (37, 13, 5, "three"), (8, 8, 4, "three")):
        if scene == "final":
            rt.upload_scene(R.final_scene_table()); rt.set_camera(R.final_camera(W, H))
        else:
            rt.upload_scene(R.three_sphere_table()); rt.set_camera(R.three_camera(W, H))
        for mode in (R.MODE_MEGAKERNEL, R.MODE_BVH):
            p = rt.params(W, H, S, seed=31, mode=mode)
            _, want, rays, _ = rt.render(p)
            want_counters = rt.counters()
            for n in (1, 2, 3, 8):
                img, total = _assemble_tile_shares(rt, p, n)
                assert total == rays, (W, H, mode, n)
                assert np.array_equal(img.cpu().numpy(), want), (W, H, mode, n)
            # counters of a lone share (n = 1) are those of the whole image
            t = torch.zeros((rt.owned_tiles(p, 0, 1), 64, 3), dtype=torch.int32, device="cuda")
            rt.render_tiles_device(p, 0, 1, d_irgb_tiles=t.data_ptr())
            assert rt.counters() == want_counters
    # float output and ids of a share; bad arguments
    rt.upload_scene(R.final_scene_table()); rt.set_camera(R.final_camera(104, 60))
    p = rt.params(104, 60, 3, seed=31)
    rgb, _, _, _ = rt.render(p)
    n = rt.owned_tiles(p, 1, 2)
    f = torch.zeros((n, 64, 3), dtype=torch.float32, device="cuda"); ids = torch.zeros(n, dtype=torch.int32, device="cuda")
    rt.render_tiles_device(p, 1, 2, d_rgb_mean_tiles=f.data_ptr(), d_tile_ids=ids.data_ptr())
    rt.stats()
    fh, ih = f.cpu().numpy(), ids.cpu().numpy()
    for k in range(n):
        ty, tx = divmod(int(ih[k]), 13)
        block = np.zeros((8, 8, 3), np.float32)
        part = rgb[ty * 8:ty * 8 + 8, tx * 8:tx * 8 + 8]
        block[:part.shape[0], :part.shape[1]] = part
        assert same(fh[k].reshape(8, 8, 3), block), k
    for first, step in ((2, 2), (-1, 2), (0, 0)):
        with pytest.raises(R.RtError):
            rt.render_tiles_device(p, first, step)
    with pytest.raises(R.RtError):
        rt.render_tiles_device(rt.params(104, 60, 3, band_rows=8, band_first=1, band_step=2), 0, 2)
    with pytest.raises(R.RtError):
        rt.render_tiles_device(rt.params(104, 60, 3, mode=R.MODE_WAVEFRONT), 0, 2)


def test_config3_eight_tile_shares_equal_the_whole_image(rt):
    """Config 3 (1200x800, 500 samples per pixel) as its eight tile shares: same integer image, same ray count."""
    import torch
    W, H, S = 1200, 800, 500
    rt.upload_scene(R.final_scene_table())
    rt.set_camera(R.final_camera(W, H))
    whole = torch.zeros((H, W, 3), dtype=torch.int32, device="cuda")
    rt.render_device(rt.params(W, H, S, seed=0), 0, whole.data_ptr())
    rays, _ = rt.stats()
    img, total = _assemble_tile_shares(rt, rt.params(W, H, S, seed=0), 8)
    assert total == rays and torch.equal(img, whole)
