"""The CPU oracle against vectors produced by the reference's own compiled classes.

Every fixture under tests/golden/ was written by oracle/ref_harness.cpp, which includes the
unmodified reference sources and is built with the same no-contraction flags as the oracle, so
equality here is bit-for-bit unless a test says otherwise.
"""
import ctypes as C
import os

import numpy as np
import pytest

import _oracle as O

G = O.GOLDEN


def _rows(name):
    out = []
    with open(os.path.join(G, name)) as f:
        for line in f:
            if line.startswith("#") or not line.strip():
                continue
            out.append(line.split())
    return out


def f32(x):
    return np.float32(float(x))


def same(a, b):
    """Bitwise float32 equality (NaN == NaN, +0 != -0 is tolerated as equal)."""
    a = np.asarray(a, dtype=np.float32)
    b = np.asarray(b, dtype=np.float32)
    return bool(np.all((a == b) | (np.isnan(a) & np.isnan(b))))


def test_philox_known_answers():
    # Random123 kat_vectors, philox4x32-10
    kats = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
            ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
            ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
             (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kats:
        c = (C.c_uint32 * 4)(*ctr)
        k = (C.c_uint32 * 2)(*key)
        o = (C.c_uint32 * 4)()
        O.lib().orc_philox4x32_10(c, k, o)
        assert tuple(o) == want


def test_sphere_hit_matches_reference():
    rows = _rows("sphere_hit.txt")
    assert len(rows) >= 1000
    hits = 0
    for r in rows:
        v = [f32(x) for x in r]
        ray = O.Ray(O.v3(v[0:3]), O.v3(v[3:6]))
        rec = O.Hit()
        got = O.lib().orc_sphere_hit(ray, O.v3(v[6:9]), v[9], v[10], v[11], C.byref(rec))
        assert got == int(r[12])
        if got:
            hits += 1
            assert same([rec.t, *rec.p.tup(), *rec.n.tup()], v[13:20])
    assert 0.3 * len(rows) < hits < 0.95 * len(rows)


def test_helpers_match_reference():
    for r in _rows("helpers.txt"):
        v = [f32(x) for x in r]
        vin, n = O.v3(v[0:3]), O.v3(v[3:6])
        assert same(O.lib().orc_reflect(vin, n).tup(), v[9:12])
        out = O.Vec3()
        ok = O.lib().orc_refract(vin, n, v[6], C.byref(out))
        assert ok == int(r[12])
        if ok:
            assert same(out.tup(), v[13:16])
        assert same([O.lib().orc_schlick(v[7], v[8])], [v[16]])


def test_scatter_matches_reference():
    rows = _rows("scatter.txt")
    seen = {(t, k): 0 for t in range(3) for k in range(2)}
    for r in rows:
        typ = int(r[0])
        v = [f32(x) for x in r[1:18]]
        albedo, param = v[0:3], v[3]
        ray = O.Ray(O.v3(v[4:7]), O.v3(v[7:10]))
        rec = O.Hit(v[10], O.v3(v[11:14]), O.v3(v[14:17]), 0)
        draws = [int(x) for x in r[18:18 + 48]]
        tail = r[18 + 48:]
        want_ret = int(tail[0])
        want = [f32(x) for x in tail[1:10]]
        want_used = int(tail[10])
        stored_param = f32(tail[11])
        if typ == 1:
            assert same([O.lib().orc_metal_fuzz(param)], [stored_param])
            param = stored_param
        u = O.usrc_replay(draws)
        att, out = O.Vec3(), O.Ray()
        ret = O.lib().orc_scatter(typ, O.v3(albedo), param, ray, C.byref(rec), C.byref(u), C.byref(att), C.byref(out))
        assert ret == want_ret
        assert u.consumed == want_used
        assert same([*att.tup(), *out.o.tup(), *out.d.tup()], want)
        seen[(typ, ret)] += 1
    # every material exercised; metal absorbs in some cases; the others always scatter
    assert seen[(0, 1)] > 100 and seen[(1, 1)] > 50 and seen[(1, 0)] > 10 and seen[(2, 1)] > 100
    assert seen[(0, 0)] == 0 and seen[(2, 0)] == 0


def test_camera_matches_reference():
    n_ray = 0
    for r in _rows("camera.txt"):
        kind, v = r[0], [f32(x) for x in r[1:]]
        cam = O.make_camera(v[0:3], v[3:6], v[6:9], v[9], v[10], v[11], v[12])
        if kind == "ctor":
            got = [*cam.origin.tup(), *cam.llc.tup(), *cam.horiz.tup(), *cam.vert.tup(), *cam.u.tup(), *cam.v.tup(),
                   *cam.w.tup(), cam.lens_radius]
            assert same(got, v[13:35])
        else:
            s, t = v[13], v[14]
            want_o, want_d = np.array(v[15:18], np.float32), np.array(v[18:21], np.float32)
            if cam.lens_radius == 0.0:
                ray = O.lib().orc_camera_get_ray(C.byref(cam), s, t, 0.0, 0.0)
                assert same([*ray.o.tup(), *ray.d.tup()], [*want_o, *want_d])
            else:
                # the reference's lens generator cannot be seeded: recover its disk sample from
                # the returned origin (u, v are orthonormal) and check the whole ray to 1e-5
                off = want_o.astype(np.float64) - np.array(cam.origin.tup())
                dx = float(np.dot(off, cam.u.tup())) / cam.lens_radius
                dy = float(np.dot(off, cam.v.tup())) / cam.lens_radius
                assert dx * dx + dy * dy < 1.0 + 1e-4
                ray = O.lib().orc_camera_get_ray(C.byref(cam), s, t, dx, dy)
                np.testing.assert_allclose([*ray.o.tup(), *ray.d.tup()], [*want_o, *want_d], rtol=1e-5, atol=2e-6)
            n_ray += 1
    assert n_ray >= 100


def test_color_paths_match_reference():
    scene = O.SceneArrays.from_table(O.load_scene_table())
    depths = []
    for r in _rows("color_paths.txt"):
        v = [f32(x) for x in r[0:9]]
        want_rays, ndraws = int(r[9]), int(r[10])
        draws = [int(x) for x in r[11:11 + ndraws]]
        u = O.usrc_replay(draws)
        rays = C.c_uint64(0)
        col = O.lib().orc_color(C.byref(scene.c), O.Ray(O.v3(v[0:3]), O.v3(v[3:6])), 0, C.byref(u), C.byref(rays))
        assert rays.value == want_rays
        assert u.consumed == ndraws
        assert same(col.tup(), v[6:9])
        depths.append(want_rays)
    assert max(depths) >= 8 and min(depths) == 1


def test_random_scene_matches_reference():
    """RandomScene() replayed from glibc's seed-1 stream reproduces the committed 487-sphere table."""
    import ctypes.util
    libc = C.CDLL(ctypes.util.find_library("c"))
    libc.srandom(1)
    u = O.usrc_libc()
    f = lambda: np.zeros(501, dtype=np.float32)
    cx, cy, cz, rad, ar, ag, ab, par = f(), f(), f(), f(), f(), f(), f(), f()
    typ = np.zeros(501, dtype=np.int32)
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    n = O.lib().orc_random_scene(C.byref(u), fp(cx), fp(cy), fp(cz), fp(rad), typ.ctypes.data_as(C.POINTER(C.c_int)),
                                 fp(ar), fp(ag), fp(ab), fp(par))
    table = O.load_scene_table()
    assert n == len(table) == 487
    assert u.consumed == 4056
    got = np.stack([cx[:n], cy[:n], cz[:n], rad[:n], typ[:n].astype(np.float32), ar[:n], ag[:n], ab[:n], par[:n]], axis=1)
    assert same(got, table.astype(np.float32))


def _meta():
    import json
    with open(os.path.join(G, "images.json")) as f:
        return json.load(f)


def _scene_cam(m):
    if m["scene"] == "final":
        ap = 0.1 if m["aperture"] < 0 else m["aperture"]
        return O.SceneArrays.from_table(O.load_scene_table()), O.final_camera(m["nx"], m["ny"], ap)
    return O.SceneArrays.from_table(O.three_sphere_table(m["scene"] == "three_hollow")), O.three_camera(m["nx"], m["ny"])


@pytest.mark.parametrize("name", ["philox_final_96x64x4", "philox_final_40x30x16_ap0",
                                  "philox_three_hollow_80x45x8", "philox_three_64x36x32"])
def test_keyed_render_is_bit_identical_to_reference(name):
    """Whole images: the reference's Color()/Sphere/Material code driven by the keyed Philox stream."""
    m = _meta()[name]
    want = np.load(os.path.join(G, f"img_{name}.npy"))
    scene, cam = _scene_cam(m)
    rgb, irgb, rays, counters = O.render_counted(scene, cam, m["nx"], m["ny"], m["ns"], O.usrc_philox(m["seed"]))
    assert rays == m["rays"]
    # RenderStats of main_modern.cpp:23-61: Color() calls, calls that hit, calls that scattered
    assert counters == (m["rays"], m["hits"], m["scatters"])
    assert same(rgb, want)
    assert np.isfinite(rgb).all()


def test_libc_render_is_bit_identical_to_reference():
    """main()'s own sequence: srand default, RandomScene(), then the pixel loop on the same rand() stream."""
    import ctypes.util
    m = _meta()["libc_final_48x32x3_ap0"]
    want = np.load(os.path.join(G, "img_libc_final_48x32x3_ap0.npy"))
    libc = C.CDLL(ctypes.util.find_library("c"))
    libc.srandom(1)
    u = O.usrc_libc()
    for _ in range(4056):          # draws RandomScene() consumed (scene_final.txt header)
        O.lib().orc_uniform(C.byref(u))
    scene, cam = _scene_cam(m)
    rgb, irgb, rays = O.render(scene, cam, m["nx"], m["ny"], m["ns"], u)
    assert rays == m["rays"]
    assert same(rgb, want)
    assert irgb.min() >= 0 and irgb.max() <= 255


def test_band_rendering_is_consistent():
    """Rows rendered in separate bands equal the same rows of a whole-image render (keyed stream)."""
    scene = O.SceneArrays.from_table(O.three_sphere_table())
    cam = O.three_camera(32, 18)
    whole, _, rays = O.render(scene, cam, 32, 18, 4, O.usrc_philox(5))
    lo, _, r0 = O.render(scene, cam, 32, 18, 4, O.usrc_philox(5), 0, 7)
    hi, _, r1 = O.render(scene, cam, 32, 18, 4, O.usrc_philox(5), 7, 18)
    assert same(np.concatenate([hi, lo], axis=0), whole) and r0 + r1 == rays


def test_get_ray_with_the_lens_matches_reference():
    """Camera::GetRay with aperture > 0 (camera.cpp:31-53): rays returned by the reference's own Camera objects with
    std::random_device pinned (oracle/ref_harness.cpp), each with the unit-disk sample its lens generator drew."""
    rows = _rows("camera_dof.txt")
    assert len(rows) >= 200
    for r in rows:
        v = [f32(x) for x in r[1:]]
        cam = O.make_camera(v[0:3], v[3:6], v[6:9], v[9], v[10], v[11], v[12])
        assert cam.lens_radius > 0.0
        ray = O.lib().orc_camera_get_ray(C.byref(cam), v[13], v[14], v[15], v[16])
        assert same([*ray.o.tup(), *ray.d.tup()], v[18:24])


def test_lens_generator_matches_reference():
    """The oracle's std::mt19937 / uniform_real_distribution<float> restatement replays the disk samples the reference's
    generator produced: same accepted samples after the same number of rejected tries."""
    O.lib().orc_lens_mt19937(1, 20240607)
    try:
        for r in _rows("camera_dof.txt"):
            want_x, want_y, tries = f32(r[16]), f32(r[17]), int(r[18])
            for k in range(tries):
                y = np.float32(2.0) * np.float32(O.lib().orc_lens_uniform()) - np.float32(1.0)      # the second argument draws first
                x = np.float32(2.0) * np.float32(O.lib().orc_lens_uniform()) - np.float32(1.0)
                inside = np.float32(np.float32(x * x) + np.float32(y * y)) < np.float32(1.0)
                assert inside == (k == tries - 1)
            assert same([x, y], [want_x, want_y])
    finally:
        O.lib().orc_lens_mt19937(0, 0)


def reference_main_ppm():
    import gzip
    with gzip.open(os.path.join(G, "reference_main_output_400x300x5.ppm.gz"), "rb") as f:
        return f.read()


def test_whole_main_output_ppm_is_reproduced():
    """output.ppm as the reference's unmodified main() wrote it (main.cpp:85-149: 400x300, 5 samples, glibc's default
    rand() stream, lens generator seeded with the pinned std::random_device value): the oracle replays the same
    sequence -- RandomScene(), camera, pixel loop with depth of field, gamma, quantisation -- and must arrive at every
    integer of the file."""
    import ctypes.util
    text = reference_main_ppm()
    assert text.startswith(b"P3\n400 300\n255\n")
    want = np.array(text.split()[4:], dtype=np.int64).reshape(300, 400, 3)
    libc = C.CDLL(ctypes.util.find_library("c"))
    libc.srandom(1)
    u = O.usrc_libc()
    f = lambda: np.zeros(501, dtype=np.float32)
    cx, cy, cz, rad, ar, ag, ab, par = f(), f(), f(), f(), f(), f(), f(), f()
    typ = np.zeros(501, dtype=np.int32)
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    n = O.lib().orc_random_scene(C.byref(u), fp(cx), fp(cy), fp(cz), fp(rad), typ.ctypes.data_as(C.POINTER(C.c_int)),
                                 fp(ar), fp(ag), fp(ab), fp(par))
    scene = O.SceneArrays(cx[:n], cy[:n], cz[:n], rad[:n], typ[:n], ar[:n], ag[:n], ab[:n], par[:n])
    cam = O.final_camera(400, 300)
    O.lib().orc_lens_mt19937(1, 20240607)
    try:
        _, irgb, _ = O.render(scene, cam, 400, 300, 5, u)
    finally:
        O.lib().orc_lens_mt19937(0, 0)
    assert np.array_equal(irgb.astype(np.int64), want)
