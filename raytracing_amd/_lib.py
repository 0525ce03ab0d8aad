"""ctypes binding of librt_amd.so (include/rt_amd.h)."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# RT_AMD_LIB selects another build of the same ABI (diagnostic builds under tools/)
LIB_PATH = os.environ.get("RT_AMD_LIB") or os.path.join(_HERE, "librt_amd.so")

MODE_MEGAKERNEL, MODE_WAVEFRONT, MODE_PIXEL, MODE_BVH = 0, 1, 2, 3
LAMBERTIAN, METAL, DIELECTRIC = 0, 1, 2

# every symbol include/rt_amd.h declares
EXPORTS = ["rt_device_count", "rt_create", "rt_destroy", "rt_last_error", "rt_upload_scene", "rt_set_camera", "rt_set_camera_ray_lists",
           "rt_owned_rows", "rt_owned_pixels", "rt_owned_tiles", "rt_render_tiles_device", "rt_render_device", "rt_render_stats", "rt_render_counters", "rt_render", "rt_render_multi",
           "rt_scene_grid", "rt_build_hierarchy", "rt_format_ppm", "rt_write_ppm", "rt_write_ppm_binary", "rt_encode_ppm_device", "rt_encode_ppm_device_noalloc", "rt_write_ppm_device", "rt_test_sphere_hit", "rt_test_scatter", "rt_test_get_ray",
           "rt_test_helpers", "rt_test_tile_lists", "rt_test_philox"]


class RtError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"rt_amd error {code}: {message}")
        self.code = code


class CameraPod(C.Structure):
    """rt_camera_pod: the public fields of the reference Camera (camera.h:13-21)."""
    _fields_ = [("origin", C.c_float * 3), ("lower_left_corner", C.c_float * 3), ("horizontal", C.c_float * 3),
                ("vertical", C.c_float * 3), ("u", C.c_float * 3), ("v", C.c_float * 3), ("w", C.c_float * 3),
                ("lens_radius", C.c_float)]


class Params(C.Structure):
    """rt_params."""
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("spp", C.c_int32), ("max_depth", C.c_int32),
                ("tmin", C.c_float), ("seed", C.c_uint64), ("band_rows", C.c_int32), ("band_first", C.c_int32),
                ("band_step", C.c_int32), ("mode", C.c_int32)]


class Counters(C.Structure):
    """rt_counters: RenderStats' three counters (performance.h:34-37)."""
    _fields_ = [("rays_traced", C.c_uint64), ("intersection_tests", C.c_uint64), ("material_evaluations", C.c_uint64)]


_lib = None


def _share_hip_runtime_with_torch():
    """PyTorch wheels carry their own copy of the HIP runtime (torch/lib/libamdhip64.so, soname libamdhip64.so.7, which
    torch's libraries ask for by file name).  If librt_amd.so is loaded first it brings in the system copy under the
    same soname, torch then loads its own copy as well, and the second runtime to start finds no GPU.  Loading torch's
    copy first, when torch is installed and not yet imported, makes both sides use one runtime whatever the import
    order.  Nothing here imports torch."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
        for base in (spec.submodule_search_locations or []) if spec else []:
            path = os.path.join(base, "lib", "libamdhip64.so")
            if os.path.exists(path):
                C.CDLL(path, mode=C.RTLD_GLOBAL)
                return
    except Exception:
        pass            # no torch, or an unusual installation: the system runtime is used


def load_library():
    """Load librt_amd.so; raises if it has not been built (there is no fallback path)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RtError(-2, f"{LIB_PATH} is missing: build it with `make -C raytracing_amd/csrc` "
                          "(or __graft_entry__.build())")
    _share_hip_runtime_with_torch()
    L = C.CDLL(LIB_PATH)
    fp, ip, vp = C.POINTER(C.c_float), C.POINTER(C.c_int32), C.c_void_p
    L.rt_device_count.argtypes = [C.POINTER(C.c_int)]
    L.rt_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.rt_destroy.argtypes = [vp]
    L.rt_destroy.restype = None
    L.rt_last_error.argtypes = [vp]
    L.rt_last_error.restype = C.c_char_p
    L.rt_upload_scene.argtypes = [vp, C.c_int, fp, fp, fp, fp, ip, fp, fp]
    L.rt_set_camera.argtypes = [vp, C.POINTER(CameraPod)]
    L.rt_set_camera_ray_lists.argtypes = [vp, C.c_int]
    L.rt_owned_rows.argtypes = [C.POINTER(Params)]
    L.rt_owned_rows.restype = C.c_int64
    L.rt_owned_pixels.argtypes = [C.POINTER(Params)]
    L.rt_owned_pixels.restype = C.c_int64
    L.rt_render_device.argtypes = [vp, C.POINTER(Params), vp, vp, vp]
    L.rt_owned_tiles.argtypes = [C.POINTER(Params), C.c_int, C.c_int]
    L.rt_owned_tiles.restype = C.c_int64
    L.rt_render_tiles_device.argtypes = [vp, C.POINTER(Params), C.c_int, C.c_int, vp, vp, vp, vp]
    L.rt_render_stats.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_float)]
    L.rt_render_counters.argtypes = [vp, C.POINTER(Counters)]
    L.rt_render.argtypes = [vp, C.POINTER(Params), fp, ip, C.POINTER(C.c_uint64), C.POINTER(C.c_float)]
    L.rt_render_multi.argtypes = [C.POINTER(vp), C.c_int, C.POINTER(Params), fp, ip, C.POINTER(C.c_uint64), fp]
    L.rt_scene_grid.argtypes = [C.c_uint64, C.c_int, C.c_int, fp, C.c_int, C.POINTER(C.c_int)]
    L.rt_build_hierarchy.argtypes = [C.c_int, fp, fp, fp, fp, fp, C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_int32)]
    L.rt_format_ppm.argtypes = [ip, C.c_int, C.c_int, C.c_char_p, C.c_int64]
    L.rt_format_ppm.restype = C.c_int64
    L.rt_write_ppm.argtypes = [C.c_char_p, ip, C.c_int, C.c_int]
    L.rt_write_ppm_binary.argtypes = [C.c_char_p, ip, C.c_int, C.c_int]
    L.rt_encode_ppm_device.argtypes = [vp, vp, C.c_int, C.c_int, vp, C.c_int64, C.POINTER(C.c_int64), vp]
    L.rt_encode_ppm_device_noalloc.argtypes = [vp, vp, C.c_int, C.c_int, vp, C.c_int64, vp, vp]
    L.rt_write_ppm_device.argtypes = [vp, C.c_char_p, vp, C.c_int, C.c_int, vp]
    L.rt_test_sphere_hit.argtypes = [vp, C.c_int, fp, fp, fp, fp, fp, fp, ip, fp, fp, fp]
    L.rt_test_scatter.argtypes = [vp, C.c_int, ip, fp, fp, fp, fp, fp, fp, ip, C.c_int, ip, fp, fp, fp, ip]
    L.rt_test_get_ray.argtypes = [vp, C.c_int, C.POINTER(CameraPod), fp, fp, fp, fp, fp]
    L.rt_test_helpers.argtypes = [vp, C.c_int, fp, fp, fp, fp, fp, fp, ip, fp, fp]
    L.rt_test_tile_lists.argtypes = [vp, C.POINTER(Params), C.POINTER(C.c_uint32), C.c_int64, C.POINTER(C.c_int64)]
    L.rt_test_philox.argtypes = [vp, C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    _lib = L
    return L


def device_count():
    n = C.c_int(0)
    rc = load_library().rt_device_count(C.byref(n))
    if rc != 0:
        raise RtError(rc, load_library().rt_last_error(None).decode())
    return n.value


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


class Camera:
    """Mirror of the reference Camera (camera.h:6-22, camera.cpp:6-29): same constructor arguments,
    same eight public fields.  Construction is host-side setup, done in float32 like the reference."""

    def __init__(self, look_from, look_at, up, vfov, aspect, aperture, focus_dist):
        f = np.float32
        v = lambda a: np.asarray(a, dtype=np.float32)
        unit = lambda a: (a / f(np.sqrt(f(f(a[0] * a[0]) + f(a[1] * a[1])) + f(a[2] * a[2])))).astype(np.float32)
        cross = lambda a, b: np.array([f(a[1] * b[2]) - f(a[2] * b[1]), -(f(a[0] * b[2]) - f(a[2] * b[0])),
                                       f(a[0] * b[1]) - f(a[1] * b[0])], dtype=np.float32)
        look_from, look_at, up = v(look_from), v(look_at), v(up)
        vfov, aspect, aperture, focus_dist = f(vfov), f(aspect), f(aperture), f(focus_dist)
        self.m_fLensRadius = f(aperture / f(2))
        theta = f(f(vfov * f(np.pi)) / f(180))
        half_h = np.tan(f(theta / f(2)), dtype=np.float32)
        half_w = f(aspect * half_h)
        self.m_oOrigin = look_from
        self.m_oW = unit(look_from - look_at)
        self.m_oU = unit(cross(up, self.m_oW))
        self.m_oV = cross(self.m_oW, self.m_oU)
        self.m_oLowerLeftCorner = (((self.m_oOrigin - f(half_w * focus_dist) * self.m_oU) - f(half_h * focus_dist) * self.m_oV)
                                   - focus_dist * self.m_oW).astype(np.float32)
        self.m_oHorizontal = (f(f(f(2) * half_w) * focus_dist) * self.m_oU).astype(np.float32)
        self.m_oVertical = (f(f(f(2) * half_h) * focus_dist) * self.m_oV).astype(np.float32)

    def pod(self):
        p = CameraPod()
        for name, src in (("origin", self.m_oOrigin), ("lower_left_corner", self.m_oLowerLeftCorner),
                          ("horizontal", self.m_oHorizontal), ("vertical", self.m_oVertical), ("u", self.m_oU),
                          ("v", self.m_oV), ("w", self.m_oW)):
            getattr(p, name)[:] = [float(x) for x in src]
        p.lens_radius = float(self.m_fLensRadius)
        return p


class Renderer:
    """One context on one GPU: upload a flattened sphere list and a camera, then render."""

    def __init__(self, device=0, camera_ray_lists=None):
        """camera_ray_lists: None keeps the library's default (on, unless RT_AMD_PRIMARY=sweep); False sends camera rays through the
        brute-force sweep like every other ray; True finds their first hit over per-tile sphere lists."""
        self._L = load_library()
        self._ctx = C.c_void_p()
        rc = self._L.rt_create(device, C.byref(self._ctx))
        if rc != 0:
            raise RtError(rc, self._L.rt_last_error(None).decode())
        self.device = device
        self.n_spheres = 0
        self.camera_ray_lists = None          # None: the library's default, whatever the environment made it
        if camera_ray_lists is not None:
            self.set_camera_ray_lists(camera_ray_lists)

    def set_camera_ray_lists(self, enabled):
        self._check(self._L.rt_set_camera_ray_lists(self._ctx, 1 if enabled else 0))
        self.camera_ray_lists = bool(enabled)

    def close(self):
        if getattr(self, "_ctx", None):
            self._L.rt_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise RtError(rc, self._L.rt_last_error(self._ctx).decode())

    def upload_scene(self, table):
        """table: (n, 9) rows of cx cy cz radius material albedo_r albedo_g albedo_b param."""
        t = np.asarray(table, dtype=np.float64)
        if t.ndim != 2 or t.shape[1] != 9:
            raise RtError(-1, "scene table must have shape (n, 9)")
        cx, cy, cz, rad = _f32(t[:, 0]), _f32(t[:, 1]), _f32(t[:, 2]), _f32(t[:, 3])
        mat = _i32(t[:, 4])
        alb = _f32(t[:, 5:8]).reshape(-1)
        par = _f32(t[:, 8])
        self._check(self._L.rt_upload_scene(self._ctx, len(cx), _fp(cx), _fp(cy), _fp(cz), _fp(rad), _ip(mat), _fp(alb), _fp(par)))
        self.n_spheres = len(cx)

    def set_camera(self, cam):
        pod = cam.pod() if hasattr(cam, "pod") else cam
        self._check(self._L.rt_set_camera(self._ctx, C.byref(pod)))

    @staticmethod
    def params(width, height, spp, seed=0, max_depth=50, tmin=0.001, band_rows=None, band_first=0, band_step=1,
               mode=MODE_MEGAKERNEL):
        return Params(width, height, spp, max_depth, tmin, seed, height if band_rows is None else band_rows,
                      band_first, band_step, mode)

    def owned_rows(self, p):
        return int(self._L.rt_owned_rows(C.byref(p)))

    def render(self, p):
        """Host-array form.  Returns (rgb_mean[rows,w,3] f32, irgb[rows,w,3] i32, rays, kernel_ms)."""
        rows = self.owned_rows(p)
        rgb = np.zeros((rows, p.width, 3), dtype=np.float32)
        irgb = np.zeros((rows, p.width, 3), dtype=np.int32)
        rays, ms = C.c_uint64(0), C.c_float(0)
        self._check(self._L.rt_render(self._ctx, C.byref(p), _fp(rgb), _ip(irgb), C.byref(rays), C.byref(ms)))
        return rgb, irgb, rays.value, ms.value

    def render_device(self, p, d_rgb_mean=0, d_irgb=0, stream=0):
        """Device-pointer form (integers: e.g. torch.Tensor.data_ptr(), torch stream .cuda_stream)."""
        self._check(self._L.rt_render_device(self._ctx, C.byref(p), C.c_void_p(d_rgb_mean or None),
                                             C.c_void_p(d_irgb or None), C.c_void_p(stream or None)))

    def owned_tiles(self, p, share_first, share_step):
        return int(self._L.rt_owned_tiles(C.byref(p), share_first, share_step))

    def render_tiles_device(self, p, share_first, share_step, d_rgb_mean_tiles=0, d_irgb_tiles=0, d_tile_ids=0, stream=0):
        """Tile share (share_first, share_step) of the image p describes, tile-major, with the image tile of each (device pointers)."""
        self._check(self._L.rt_render_tiles_device(self._ctx, C.byref(p), share_first, share_step, C.c_void_p(d_rgb_mean_tiles or None),
                                                   C.c_void_p(d_irgb_tiles or None), C.c_void_p(d_tile_ids or None), C.c_void_p(stream or None)))

    def stats(self):
        rays, ms = C.c_uint64(0), C.c_float(0)
        self._check(self._L.rt_render_stats(self._ctx, C.byref(rays), C.byref(ms)))
        return rays.value, ms.value

    def counters(self):
        """(rays_traced, intersection_tests, material_evaluations) of the last render."""
        c = Counters()
        self._check(self._L.rt_render_counters(self._ctx, C.byref(c)))
        return int(c.rays_traced), int(c.intersection_tests), int(c.material_evaluations)

    def encode_ppm_device(self, d_irgb, width, height, d_text=0, cap=0, stream=0):
        """P3 text of a device-resident int32 image, written to device memory.  Returns the text length;
        with d_text = 0 only the length is computed."""
        n = C.c_int64(0)
        rc = self._L.rt_encode_ppm_device(self._ctx, C.c_void_p(d_irgb), width, height, C.c_void_p(d_text or None), cap,
                                          C.byref(n), C.c_void_p(stream or None))
        if rc != 0 and not (rc == -5 and not d_text):
            self._check(rc)
        return n.value

    def encode_ppm_device_noalloc(self, d_irgb, width, height, d_text, cap, d_bytes, stream=0):
        """rt_encode_ppm_device_noalloc: enqueue only; the length arrives in the int64 at device address d_bytes."""
        self._check(self._L.rt_encode_ppm_device_noalloc(self._ctx, C.c_void_p(d_irgb or None), width, height, C.c_void_p(d_text or None), cap,
                                                         C.c_void_p(d_bytes or None), C.c_void_p(stream or None)))

    def write_ppm_device(self, path, d_irgb, width, height, stream=0):
        self._check(self._L.rt_write_ppm_device(self._ctx, os.fsencode(path), C.c_void_p(d_irgb), width, height, C.c_void_p(stream or None)))

    # ---- unit-level entry points -------------------------------------------------------
    def test_sphere_hit(self, ray_o, ray_d, center, radius, tmin, tmax):
        n = len(radius)
        ro, rd, c = _f32(ray_o).reshape(-1), _f32(ray_d).reshape(-1), _f32(center).reshape(-1)
        r, a, b = _f32(radius), _f32(tmin), _f32(tmax)
        hit = np.zeros(n, np.int32); t = np.zeros(n, np.float32); p = np.zeros(3 * n, np.float32); nr = np.zeros(3 * n, np.float32)
        self._check(self._L.rt_test_sphere_hit(self._ctx, n, _fp(ro), _fp(rd), _fp(c), _fp(r), _fp(a), _fp(b), _ip(hit), _fp(t), _fp(p), _fp(nr)))
        return hit, t, p.reshape(n, 3), nr.reshape(n, 3)

    def test_scatter(self, material, albedo, param, ray_o, ray_d, hit_p, hit_n, draws):
        n = len(param)
        m, al, pa = _i32(material), _f32(albedo).reshape(-1), _f32(param)
        ro, rd, hp, hn = (_f32(x).reshape(-1) for x in (ray_o, ray_d, hit_p, hit_n))
        dr = _i32(draws)
        per = dr.shape[1]
        dr = dr.reshape(-1)
        ret = np.zeros(n, np.int32); used = np.zeros(n, np.int32)
        att, oo, od = (np.zeros(3 * n, np.float32) for _ in range(3))
        self._check(self._L.rt_test_scatter(self._ctx, n, _ip(m), _fp(al), _fp(pa), _fp(ro), _fp(rd), _fp(hp), _fp(hn), _ip(dr), per,
                                            _ip(ret), _fp(att), _fp(oo), _fp(od), _ip(used)))
        return ret, att.reshape(n, 3), oo.reshape(n, 3), od.reshape(n, 3), used

    def test_get_ray(self, cam, s, t, disk_xy):
        n = len(s)
        pod = cam.pod() if hasattr(cam, "pod") else cam
        s_, t_, d_ = _f32(s), _f32(t), _f32(disk_xy).reshape(-1)
        ro, rd = np.zeros(3 * n, np.float32), np.zeros(3 * n, np.float32)
        self._check(self._L.rt_test_get_ray(self._ctx, n, C.byref(pod), _fp(s_), _fp(t_), _fp(d_), _fp(ro), _fp(rd)))
        return ro.reshape(n, 3), rd.reshape(n, 3)

    def test_helpers(self, v, normal, ni_over_nt, cosine, ref_idx):
        """Reflect / Refract / Schlick on arrays -> (reflected[n,3], refract_ok[n], refracted[n,3], schlick[n])."""
        n = len(cosine)
        v_, n_ = _f32(v).reshape(-1), _f32(normal).reshape(-1)
        a, b, c = _f32(ni_over_nt), _f32(cosine), _f32(ref_idx)
        refl, refr, sch = np.zeros(3 * n, np.float32), np.zeros(3 * n, np.float32), np.zeros(n, np.float32)
        ok = np.zeros(n, np.int32)
        self._check(self._L.rt_test_helpers(self._ctx, n, _fp(v_), _fp(n_), _fp(a), _fp(b), _fp(c), _fp(refl), _ip(ok), _fp(refr), _fp(sch)))
        return refl.reshape(n, 3), ok, refr.reshape(n, 3), sch

    def test_tile_lists(self, p):
        """Candidate lists of the camera rays, per 8x8 tile of the owned rows: (counts[tiles] (-1: no list), lists[tiles, 63])."""
        tiles = ((p.width + 7) // 8) * ((self.owned_rows(p) + 7) // 8)
        words = np.zeros(max(tiles, 1) * 64, dtype=np.uint32)
        got = C.c_int64(0)
        self._check(self._L.rt_test_tile_lists(self._ctx, C.byref(p), words.ctypes.data_as(C.POINTER(C.c_uint32)), words.size, C.byref(got)))
        words = words[:got.value * 64].reshape(got.value, 64)
        return np.where(words[:, 0] == 0xffffffff, -1, words[:, 0].astype(np.int64)), words[:, 1:]

    def test_philox(self, ctr, key):
        c = np.ascontiguousarray(ctr, dtype=np.uint32).reshape(-1)
        k = np.ascontiguousarray(key, dtype=np.uint32)
        n = len(c) // 4
        out = np.zeros(4 * n, np.uint32)
        u32 = C.POINTER(C.c_uint32)
        self._check(self._L.rt_test_philox(self._ctx, n, c.ctypes.data_as(u32), k.ctypes.data_as(u32), out.ctypes.data_as(u32)))
        return out.reshape(n, 4)


def render_multi(renderers, p):
    """rt_render_multi: one image over the contexts of `renderers` (each holding the same scene and camera).
    Returns (rgb_mean[h,w,3] f32, irgb[h,w,3] i32, rays, [kernel_ms per context])."""
    L = load_library()
    n = len(renderers)
    ctxs = (C.c_void_p * n)(*[r._ctx for r in renderers])
    rgb = np.zeros((p.height, p.width, 3), dtype=np.float32)
    irgb = np.zeros((p.height, p.width, 3), dtype=np.int32)
    rays = C.c_uint64(0)
    ms = np.zeros(n, dtype=np.float32)
    rc = L.rt_render_multi(ctxs, n, C.byref(p), _fp(rgb), _ip(irgb), C.byref(rays), _fp(ms))
    if rc != 0:
        raise RtError(rc, L.rt_last_error(renderers[0]._ctx).decode() if n else L.rt_last_error(None).decode())
    return rgb, irgb, rays.value, [float(x) for x in ms]


def scene_grid(seed=42, half_extent=75, max_small=19996):
    """rt_scene_grid: RandomScene()'s rule on a (2*half_extent)^2 grid with an explicit Philox draw order."""
    cap = (2 * half_extent) ** 2 + 4
    table = np.zeros((cap, 9), dtype=np.float32)
    n = C.c_int(0)
    L = load_library()
    rc = L.rt_scene_grid(seed, half_extent, max_small, _fp(table.reshape(-1)), cap, C.byref(n))
    if rc != 0:
        raise RtError(rc, L.rt_last_error(None).decode())
    return table[:n.value].astype(np.float64)


def format_ppm(irgb):
    """ASCII P3 bytes exactly as main.cpp:95-97,143-144 writes them."""
    a = _i32(irgb)
    h, w = a.shape[0], a.shape[1]
    L = load_library()
    need = -L.rt_format_ppm(_ip(a.reshape(-1)), w, h, None, 0)
    buf = C.create_string_buffer(int(need))
    n = L.rt_format_ppm(_ip(a.reshape(-1)), w, h, buf, need)
    return buf.raw[:n]


def write_ppm(path, irgb):
    a = _i32(irgb)
    rc = load_library().rt_write_ppm(os.fsencode(path), _ip(a.reshape(-1)), a.shape[1], a.shape[0])
    if rc != 0:
        raise RtError(rc, load_library().rt_last_error(None).decode())


def write_ppm_binary(path, irgb):
    """Binary P6 form of the same image (values must fit a byte)."""
    a = _i32(irgb)
    rc = load_library().rt_write_ppm_binary(os.fsencode(path), _ip(a.reshape(-1)), a.shape[1], a.shape[0])
    if rc != 0:
        raise RtError(rc, load_library().rt_last_error(None).decode())


def build_hierarchy(table):
    """rt_build_hierarchy on an (n, 9) scene table.  Returns (nodes[k, 8] float32 with the link in column 7's bits,
    leaf_index, big, depth)."""
    t = np.asarray(table, dtype=np.float64)
    n = len(t)
    cx, cy, cz, rad = _f32(t[:, 0]), _f32(t[:, 1]), _f32(t[:, 2]), _f32(t[:, 3])
    nodes = np.zeros((2 * n + 2, 8), dtype=np.float32)
    leaf = np.zeros(n + 1, dtype=np.uint32)
    big = np.zeros(16, dtype=np.uint32)
    counts = (C.c_int32 * 4)()
    L = load_library()
    u32 = C.POINTER(C.c_uint32)
    rc = L.rt_build_hierarchy(n, _fp(cx), _fp(cy), _fp(cz), _fp(rad), _fp(nodes.reshape(-1)), len(nodes), leaf.ctypes.data_as(u32),
                              big.ctypes.data_as(u32), counts)
    if rc != 0:
        raise RtError(rc, L.rt_last_error(None).decode())
    return nodes[:counts[0]], leaf[:counts[1]], big[:counts[2]], counts[3]
