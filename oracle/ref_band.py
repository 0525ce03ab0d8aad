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



_REF_EXACT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "ref_harness")


def exact_available():
    return os.path.exists(_REF_EXACT)


def keyed_band_exact(width, height, spp, j0, rows, seed):
    """Rows j0 .. j0 + rows (counted from the bottom, as main.cpp:118 does) of the final scene as the reference's own code
    computes them on the keyed random stream of the GPU path -- the pinned build without FMA contraction (oracle/Makefile),
    run as a child process.  Returns (float32 array [rows, width, 3] with the rows top to bottom, ray count)."""
    import subprocess
    import tempfile
    import numpy as np
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "band.f32")
        r = subprocess.run([_REF_EXACT, "render", "final", str(width), str(height), str(spp), "philox", str(seed), str(j0), str(j0 + rows), "-1", out],
                           capture_output=True, text=True, check=True)
        rays = int(r.stdout.split()[1])
        return np.fromfile(out, dtype=np.float32).reshape(rows, width, 3), rays
