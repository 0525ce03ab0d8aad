"""ctypes binding of the CPU oracle (oracle/libcpu_ref.so).  Test infrastructure only."""
import ctypes as C
import os
import subprocess
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
GOLDEN = os.path.join(ROOT, "tests", "golden")


class Vec3(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]

    def tup(self):
        return (self.x, self.y, self.z)


class Ray(C.Structure):
    _fields_ = [("o", Vec3), ("d", Vec3)]


class Hit(C.Structure):
    _fields_ = [("t", C.c_float), ("p", Vec3), ("n", Vec3), ("idx", C.c_int)]


class Scene(C.Structure):
    _fields_ = [("n", C.c_int)] + [(k, C.POINTER(C.c_float)) for k in ("cx", "cy", "cz", "radius")] + \
               [("type", C.POINTER(C.c_int))] + [(k, C.POINTER(C.c_float)) for k in ("ar", "ag", "ab", "param")]


class Camera(C.Structure):
    _fields_ = [(k, Vec3) for k in ("origin", "llc", "horiz", "vert", "u", "v", "w")] + [("lens_radius", C.c_float)]


class USrc(C.Structure):
    _fields_ = [("kind", C.c_int), ("replay", C.POINTER(C.c_int)), ("n_replay", C.c_long), ("pos", C.c_long),
                ("seed", C.c_uint64), ("px", C.c_uint32), ("smp", C.c_uint32), ("bounce", C.c_uint32),
                ("draw", C.c_uint32), ("consumed", C.c_long)]


def v3(a):
    return Vec3(float(a[0]), float(a[1]), float(a[2]))


_lib = None


def lib(fast=False):
    """Load the oracle; (re)build it with make when the shared object is missing."""
    global _lib
    name = "libcpu_ref_fast.so" if fast else "libcpu_ref.so"
    path = os.path.join(ORACLE_DIR, name)
    if not os.path.exists(path):
        subprocess.check_call(["make", "-C", ORACLE_DIR, name])
    if fast:
        L = C.CDLL(path)
        _proto(L)
        return L
    if _lib is None:
        _lib = C.CDLL(path)
        _proto(_lib)
    return _lib


def _proto(L):
    L.orc_philox4x32_10.argtypes = [C.POINTER(C.c_uint32)] * 3
    L.orc_uniform.restype = C.c_double
    L.orc_uniform.argtypes = [C.POINTER(USrc)]
    L.orc_sphere_hit.restype = C.c_int
    L.orc_sphere_hit.argtypes = [Ray, Vec3, C.c_float, C.c_float, C.c_float, C.POINTER(Hit)]
    L.orc_world_hit.restype = C.c_int
    L.orc_world_hit.argtypes = [C.POINTER(Scene), Ray, C.c_float, C.c_float, C.POINTER(Hit)]
    L.orc_reflect.restype = Vec3
    L.orc_reflect.argtypes = [Vec3, Vec3]
    L.orc_refract.restype = C.c_int
    L.orc_refract.argtypes = [Vec3, Vec3, C.c_float, C.POINTER(Vec3)]
    L.orc_schlick.restype = C.c_float
    L.orc_schlick.argtypes = [C.c_float, C.c_float]
    L.orc_metal_fuzz.restype = C.c_float
    L.orc_metal_fuzz.argtypes = [C.c_float]
    L.orc_scatter.restype = C.c_int
    L.orc_scatter.argtypes = [C.c_int, Vec3, C.c_float, Ray, C.POINTER(Hit), C.POINTER(USrc), C.POINTER(Vec3), C.POINTER(Ray)]
    L.orc_camera_init.argtypes = [C.POINTER(Camera), Vec3, Vec3, Vec3, C.c_float, C.c_float, C.c_float, C.c_float]
    L.orc_camera_get_ray.restype = Ray
    L.orc_camera_get_ray.argtypes = [C.POINTER(Camera), C.c_float, C.c_float, C.c_float, C.c_float]
    L.orc_color.restype = Vec3
    L.orc_color.argtypes = [C.POINTER(Scene), Ray, C.c_int, C.POINTER(USrc), C.POINTER(C.c_uint64)]
    L.orc_render.restype = C.c_uint64
    L.orc_render.argtypes = [C.POINTER(Scene), C.POINTER(Camera), C.c_int, C.c_int, C.c_int, C.POINTER(USrc),
                             C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_int)]
    L.orc_random_scene.restype = C.c_int
    L.orc_random_scene.argtypes = [C.POINTER(USrc)] + [C.POINTER(C.c_float)] * 4 + [C.POINTER(C.c_int)] + [C.POINTER(C.c_float)] * 4
    L.orc_quantise.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_int)]
    L.orc_counters_reset.argtypes = []
    L.orc_lens_mt19937.argtypes = [C.c_int, C.c_uint32]
    L.orc_lens_uniform.restype = C.c_float
    L.orc_lens_uniform.argtypes = []
    L.orc_counters_get.argtypes = [C.POINTER(C.c_uint64)]


class SceneArrays:
    """Owns the numpy arrays behind an orc_scene."""

    def __init__(self, cx, cy, cz, radius, type_, ar, ag, ab, param):
        f = lambda a: np.ascontiguousarray(a, dtype=np.float32)
        self.cx, self.cy, self.cz, self.radius = f(cx), f(cy), f(cz), f(radius)
        self.type = np.ascontiguousarray(type_, dtype=np.int32)
        self.ar, self.ag, self.ab, self.param = f(ar), f(ag), f(ab), f(param)
        self.n = len(self.cx)
        fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
        self.c = Scene(self.n, fp(self.cx), fp(self.cy), fp(self.cz), fp(self.radius),
                       self.type.ctypes.data_as(C.POINTER(C.c_int)), fp(self.ar), fp(self.ag), fp(self.ab), fp(self.param))

    @staticmethod
    def from_table(t):
        t = np.asarray(t, dtype=np.float64)
        return SceneArrays(t[:, 0], t[:, 1], t[:, 2], t[:, 3], t[:, 4].astype(np.int32), t[:, 5], t[:, 6], t[:, 7], t[:, 8])


def load_scene_table(name="scene_final.txt"):
    return np.loadtxt(os.path.join(GOLDEN, name), comments="#", dtype=np.float64, ndmin=2)


def three_sphere_table(hollow=False):
    """Four-sphere material scene (performance_comparison.cpp:48-51); optional hollow glass shell (main.cpp:107)."""
    rows = [[0.0, -100.5, -1.0, 100.0, 0, 0.5, 0.5, 0.5, 0.0],
            [0.0, 0.0, -1.0, 0.5, 0, 0.7, 0.3, 0.3, 0.0],
            [-1.0, 0.0, -1.0, 0.5, 2, 1.0, 1.0, 1.0, 1.5],
            [1.0, 0.0, -1.0, 0.5, 1, 0.8, 0.6, 0.2, 0.3]]
    if hollow:
        rows.append([-1.0, 0.0, -1.0, -0.45, 2, 1.0, 1.0, 1.0, 1.5])
    return np.array(rows, dtype=np.float64)


def make_camera(from_, at, up, vfov, aspect, aperture, focus):
    cam = Camera()
    lib().orc_camera_init(C.byref(cam), v3(from_), v3(at), v3(up), vfov, aspect, aperture, focus)
    return cam


def final_camera(nx, ny, aperture=0.1):
    return make_camera((13, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, np.float32(nx) / np.float32(ny), aperture, 10.0)


def three_camera(nx, ny):
    return make_camera((0, 0, 0), (0, 0, -1), (0, 1, 0), 90.0, np.float32(nx) / np.float32(ny), 0.0, 1.0)


def usrc_replay(draws):
    arr = np.ascontiguousarray(draws, dtype=np.int32)
    u = USrc(0, arr.ctypes.data_as(C.POINTER(C.c_int)), len(arr), 0, 0, 0, 0, 0, 0, 0)
    u._keep = arr
    return u


def usrc_philox(seed):
    return USrc(2, None, 0, 0, seed, 0, 0, 0, 0, 0)


def usrc_libc():
    return USrc(1, None, 0, 0, 0, 0, 0, 0, 0, 0)


def render(scene, cam, nx, ny, ns, usrc, j0=0, j1=None, fast=False):
    """Returns (rgb_mean[rows,nx,3] float32, irgb int32, rays)."""
    j1 = ny if j1 is None else j1
    rgb = np.zeros((j1 - j0, nx, 3), dtype=np.float32)
    irgb = np.zeros((j1 - j0, nx, 3), dtype=np.int32)
    rays = lib(fast).orc_render(C.byref(scene.c), C.byref(cam), nx, ny, ns, C.byref(usrc), j0, j1,
                                rgb.ctypes.data_as(C.POINTER(C.c_float)), irgb.ctypes.data_as(C.POINTER(C.c_int)))
    return rgb, irgb, int(rays)


def render_counted(scene, cam, nx, ny, ns, usrc, j0=0, j1=None):
    """render() plus the RenderStats counters (rays_traced, intersection_tests, material_evaluations)."""
    L = lib()
    L.orc_counters_reset()
    rgb, irgb, rays = render(scene, cam, nx, ny, ns, usrc, j0, j1)
    c = (C.c_uint64 * 3)()
    L.orc_counters_get(c)
    return rgb, irgb, rays, (int(c[0]), int(c[1]), int(c[2]))
