"""TEST INFRASTRUCTURE — CPU-baseline worker: one process renders one band of rows with the reference build
(oracle/_ref/libref_render.so).  Imported by bench.py's cpu_baseline leg only; kept free of torch so that a
process pool starts quickly."""
import ctypes as C
import os
import time

_REF = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "libref_render.so")


def available():
    return os.path.exists(_REF)


def render_band(args):
    """args = (width, height, spp, j0, rows, seed) -> (rays, seconds).  One call = one core, own rand() stream."""
    width, height, spp, j0, rows, seed = args
    L = C.CDLL(_REF)
    L.ref_render_rows.restype = C.c_uint64
    L.ref_render_rows.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_uint, C.c_int, C.c_int, C.POINTER(C.c_float)]
    t0 = time.perf_counter()
    rays = L.ref_render_rows(b"final", width, height, spp, seed, j0, j0 + rows, None)
    return int(rays), time.perf_counter() - t0
