#!/usr/bin/env python3
"""Throughput of the render loop on MI355X: Mrays/s on the 1200x800 RTOW final scene.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c3|c1|c2|c4|c5]

The same command line works for every N.  With N > 1 and no rank environment the script starts N worker
processes itself (`python -m torch.distributed.run`, one per GPU) before anything in it touches a GPU and
relays their output; under an existing launcher (RANK / WORLD_SIZE set) it is one of the ranks.

One step = one render of the whole image: each rank renders its interleaved row bands (scene and camera
replicated, resident in HBM before the timed region) and, for N > 1, one RCCL gather assembles the rows on
rank 0.  Total work is fixed as N grows ("strong" scaling).  Rank 0 prints one JSON line.

The timed path is the brute-force one: EVERY ray, camera rays included, is tested against every sphere
(HittableList::Hit, hittablelist.cpp:21-29, over Sphere::Hit, sphere.cpp:15-52), so `value` and `roofline`
describe that loop.  The library's shortcut for camera rays (per-tile sphere lists, rt_set_camera_ray_lists)
is measured beside it, outside the timed region, as the `beyond_brute_force` object of the line.
"""
import argparse
import ctypes as C
import glob
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import raytracing_amd as R                                   # noqa: E402
from raytracing_amd import dist as rdist                     # noqa: E402

PEAK_FP32_VALU_TFLOPS = 157.3      # 256 CU x 4 SIMD x 32 lanes x 2 flop (FMA) x 2.4 GHz
# plain v_fma_f32 chains held for 0.1-1 s on this part (tools/ubench_sustain.hip); the plain FMA runs at the full rate, so the
# 78.6 TFLOP/s "non-packed" figure quoted for older parts is not a ceiling here
SUSTAINED_FP32_FMA_TFLOPS = 148.0
FLOP_PER_TEST = 18                 # oc (3) + oc.d (5) + oc.oc (5) + -r^2 (1) + b^2 (1) + a*c (1) + - (1) + compare (1)
FLOP_PER_RAY_EXTRA = 150           # hit resolve + scatter, per closest-hit query

WORKLOADS = {
    # name: (scene, width, height, spp)
    "c1": ("final", 1200, 800, 10),
    "c2": ("three", 400, 225, 100),
    "c3": ("final", 1200, 800, 500),
    "c4": ("final", 3840, 2160, 1024),
    "c5": ("grid20k", 1920, 1080, 256),
}


def profiled_counters():
    """Per-ray figures of k_trace from the newest rocprofv3 counter summary under profiles/ (written by
    tools/pmc_summary.py from the passes of tools/profile_round.sh on the default workload): HBM bytes, vector
    instructions, executed FP32 operations.  None when no summary is present."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_k_trace.json")))
    if not files:
        return None
    with open(files[-1]) as f:
        d = json.load(f)
    d["file"] = os.path.relpath(files[-1], ROOT)
    return d


def cpu_baseline(width, height, spp):
    """The reference's own compiled render code (oracle/_ref, built from /root/reference by
    oracle/Makefile) on one host core, on a bounded band of the same workload.  Falls back to the
    C port of the oracle when the reference build is not present."""
    rows = 40
    j0 = height // 2 - rows // 2
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ref_band
    if ref_band.available():
        rays, dt = ref_band.render_band((width, height, spp, j0, rows, 1))
        kind = "reference"
    else:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import _oracle as O
        scene = O.SceneArrays.from_table(R.final_scene_table())
        cam = O.final_camera(width, height)
        t0 = time.perf_counter()
        _, _, rays = O.render(scene, cam, width, height, spp, O.usrc_philox(0), j0, j0 + rows, fast=True)
        dt = time.perf_counter() - t0
        kind = "port"
    return {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": 1, "kind": kind,
            "sample": f"rows {j0}..{j0 + rows} of {width}x{height}, {spp} spp, final scene: {rays} rays in {dt:.1f} s",
            "samples_per_s": rows * width * spp / dt}


def cpu_baseline_multicore(width, height, spp, cores):
    """Pixel-parallel variant (what main_modern.cpp:172-195 does with OpenMP, but with thread-private random
    streams): `cores` processes, each rendering its own 24-row band of the same image at the same time."""
    import multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ref_band
    if not ref_band.available():
        return None
    rows = 24
    bands = [(width, height, spp, (height // cores) * k, rows, 100 + k) for k in range(cores)]
    t0 = time.perf_counter()
    with mp.get_context("spawn").Pool(cores) as pool:
        res = pool.map(ref_band.render_band, bands)
    dt = time.perf_counter() - t0
    busy = max(r[1] for r in res)
    rays = sum(r[0] for r in res)
    return {"value": rays / busy / 1e6, "unit": "Mrays/s", "cores": cores, "kind": "reference",
            "sample": f"{cores} processes x {rows} rows of {width}x{height}, {spp} spp, final scene: {rays} rays, slowest band {busy:.1f} s (pool wall {dt:.1f} s)"}


def ppm_encode_report(rt, img):
    """The step after the render loop (output.ppm, main.cpp:91-97,143-147), outside the timed region: ASCII P3 text
    of the finished image produced on the GPU (rt_encode_ppm_device), next to the single-pass host formatter."""
    h, w = img.shape[0], img.shape[1]
    img = img.contiguous()
    need = rt.encode_ppm_device(img.data_ptr(), w, h)
    text = torch.empty(need + 8, dtype=torch.uint8, device=img.device)
    stream = torch.cuda.current_stream().cuda_stream
    rt.encode_ppm_device(img.data_ptr(), w, h, text.data_ptr(), need, stream)
    torch.cuda.synchronize()
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps):
        rt.encode_ppm_device(img.data_ptr(), w, h, text.data_ptr(), need, stream)
    torch.cuda.synchronize()
    gpu_ms = (time.perf_counter() - t0) / reps * 1e3
    pinned = torch.empty(need, dtype=torch.uint8, pin_memory=True)
    t0 = time.perf_counter()
    pinned.copy_(text[:need]); torch.cuda.synchronize()
    copy_ms = (time.perf_counter() - t0) * 1e3
    host_img = img.cpu().numpy().reshape(-1)
    L = R.load_library()
    cap = 32 + w * h * 36
    buf = (C.c_char * cap)()
    t0 = time.perf_counter()
    n = L.rt_format_ppm(host_img.ctypes.data_as(C.POINTER(C.c_int32)), w, h, buf, cap)
    host_ms = (time.perf_counter() - t0) * 1e3
    same = bytes(buf[:n]) == pinned.numpy().tobytes()
    return {"text_bytes": need, "gpu_encode_ms": gpu_ms, "text_copy_to_host_ms": copy_ms, "host_formatter_ms": host_ms,
            "identical": bool(same), "note": "three kernels (length, scan, write) including the host wait for the length; not part of the timed region"}


def parse_args(argv):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--spp", type=int, default=0, help="override samples per pixel")
    ap.add_argument("--no-beyond", "--no-sweep-only", dest="no_beyond", action="store_true",
                    help="skip the extra renders with camera rays through per-tile sphere lists (the beyond_brute_force object of the line)")
    ap.add_argument("--no-config1", action="store_true", help="skip the extra renders of the short launch (the config1 object of the line)")
    ap.add_argument("--camera-ray-lists", action="store_true", help="time the path with the camera-ray shortcut instead of the brute-force one (not the headline)")
    ap.add_argument("--force-dist", action="store_true", help="N = 1: still create the RCCL process group (one rank) and run the gather and the reductions through it")
    ap.add_argument("--band-rows", type=int, default=0, help="rows per band (0 = 8 on one GPU, 1 on several)")
    ap.add_argument("--partition", default="rows", choices=["tiles", "rows"],
                    help="how one image is dealt to several GPUs: interleaved row bands (default; single rows measure best: the slowest of 8 shares of "
                         "config 3 takes 10.4 ms on one GPU against 11.2 ms for tile shares), or every N-th 8x8 tile of the library's work order")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--mode", default="megakernel", choices=["megakernel", "wavefront", "pixel", "bvh"])
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="debug: all ranks share GPU 0 and gather over gloo (RCCL refuses two ranks on one device)")
    ap.add_argument("--no-overlap", action="store_true",
                    help="several GPUs, row bands: wait for every step's gather before the next step's kernels start (default: the gather of "
                         "step k runs on the collective's stream while step k + 1 renders into a second buffer)")
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU: the ranks start, gather synthetic rows over gloo and rank 0 checks the assembled image (launcher, "
                         "partition and gather without the render kernels)")
    return ap.parse_args(argv)


def self_launch(args, argv):
    """N > 1 without a rank environment: start the ranks as children of this process, which has not touched a GPU (importing
    torch does not), wait for them and pass their exit status on.  Never an exec: the parent stays a plain relay."""
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=env)


def dry_run(args, rank, world):
    """Launcher, partition and gather on the CPU: every rank fills its rows with a function of the image row and rank 0
    checks the gathered image."""
    _, width, height, _ = WORKLOADS[args.workload]
    band = args.band_rows if args.band_rows > 0 else (8 if world == 1 else 1)
    if world > 1:
        dist.init_process_group("gloo")
    if args.partition == "tiles":
        # every rank fills its share of a made-up work order (a fixed permutation of the tiles) with a function of the tile
        tx, ty = rdist.tile_grid(width, height)
        order = np.random.default_rng(1).permutation(tx * ty)
        mine = order[rank::world]
        g = rdist.TileGather(width, height, rank, world)
        tiles = torch.zeros((g.pad, 64, 3), dtype=torch.int32)
        ids = torch.zeros(g.pad, dtype=torch.int32)
        ids[:len(mine)] = torch.from_numpy(mine.astype(np.int32))
        tiles[:len(mine)] = (ids[:len(mine), None, None] * 7 + torch.arange(64, dtype=torch.int32)[None, :, None] + torch.arange(3, dtype=torch.int32)[None, None, :]) % 256
        t0 = time.perf_counter()
        g.set_ids(ids)
        img = g.gather(tiles)
        gather_ms = (time.perf_counter() - t0) * 1e3
        if rank == 0:
            yy, xx = torch.meshgrid(torch.arange(height), torch.arange(width), indexing="ij")
            tid = (yy // 8) * tx + xx // 8
            l = (yy % 8) * 8 + xx % 8
            want = ((tid[..., None] * 7 + l[..., None] + torch.arange(3)[None, None, :]) % 256).to(torch.int32)
            ok = bool(torch.equal(img, want))
            emit({"metric": "Mrays/s (1200x800 RTOW final scene)", "value": None, "unit": "Mrays/s", "n_gpus": world, "dry_run": True,
                              "ranks_reported_by_backend": dist.get_world_size() if world > 1 else 1, "backend": "gloo" if world > 1 else None,
                              "assembled_image_ok": ok, "gather_ms": gather_ms, "config": {"workload": args.workload, "partition": "tiles"}})
            if not ok:
                raise SystemExit("dry run: the gathered image is not the expected one")
        if world > 1:
            dist.destroy_process_group()
        return
    pad = rdist.max_owned_rows(height, band, world)
    local = torch.zeros((pad, width, 3), dtype=torch.int32)
    k = 0
    for y0, y1 in rdist.owned_bands(height, band, rank, world):
        rows = torch.arange(y0, y1, dtype=torch.int32)
        local[k:k + (y1 - y0)] = (rows[:, None, None] * 7 + torch.arange(3, dtype=torch.int32)[None, None, :]) % 256
        k += y1 - y0
    t0 = time.perf_counter()
    img = rdist.gather_rows(local, height, band, rank, world)
    gather_ms = (time.perf_counter() - t0) * 1e3
    if rank == 0:
        want = (torch.arange(height, dtype=torch.int32)[:, None, None] * 7 + torch.arange(3, dtype=torch.int32)[None, None, :]) % 256
        ok = bool(torch.equal(img, want.expand(height, width, 3)))
        emit({"metric": "Mrays/s (1200x800 RTOW final scene)", "value": None, "unit": "Mrays/s", "n_gpus": world, "dry_run": True,
                          "ranks_reported_by_backend": dist.get_world_size() if world > 1 else 1, "backend": "gloo" if world > 1 else None,
                          "assembled_image_ok": ok, "gather_ms": gather_ms, "config": {"workload": args.workload, "band_rows": band}})
        if not ok:
            raise SystemExit("dry run: the gathered image is not the expected one")
    if world > 1:
        dist.destroy_process_group()


def setup_scene(rt, scene_name, width, height):
    if scene_name == "final":
        table = R.final_scene_table()
        cam = R.final_camera(width, height)
    elif scene_name == "grid20k":
        table = R.scene_grid(42, 75, 19996)
        cam = R.final_camera(width, height)
    else:
        table = R.three_sphere_table()
        cam = R.three_camera(width, height)
    rt.upload_scene(table)
    rt.set_camera(cam)
    return len(table)


def roofline_block(args, n_spheres, rays_per_launch, kernel_ms, prof):
    """The FP32 vector-ALU roofline of the dominant kernel, k_trace on the brute-force path: every ray against every sphere.
    `achieved` prices each ray-sphere test at the reference's 18 flop (sphere.cpp:17-25) plus 150 flop per ray for hit record
    and scatter: rays x (18 x spheres + 150) / kernel time (the time of ALL kernels of the step: camera rays, k_trace, resolve).
    The kernel makes every one of those tests, but as a conservative filter (the pair's dot product on the bf16 matrix pipe,
    verdict and accumulate on the vector ALU) followed by the reference's own arithmetic on the few survivors, so it executes
    fewer FP32 operations than it is credited with -- `executed_*`, from the profiled instruction counters, is what the vector
    ALU did -- and, since the dot products do not run on the pipe whose peak is the denominator, `frac` can pass 1."""
    flop_per_ray = FLOP_PER_TEST * n_spheres + FLOP_PER_RAY_EXTRA
    achieved = rays_per_launch * flop_per_ray / (kernel_ms * 1e-3) / 1e12
    block = {"bound": "valu", "achieved": achieved, "peak": PEAK_FP32_VALU_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_FP32_VALU_TFLOPS,
             "frac_is": "rays x (18 x spheres + 150) flop per launch / time of all kernels of the step / FP32 vector peak, every ray tested against every "
                        "sphere (brute force).  Above 1 means faster than a literal FP32 vector-ALU implementation could be at its peak: the "
                        "18-flop test's dot product runs on the bf16 matrix pipe, which that peak does not count.  executed_frac is what the "
                        "vector ALU executed, valu_issue_frac how busy its issue port was",
             "peak_sustained_measured": SUSTAINED_FP32_FMA_TFLOPS, "frac_of_sustained": achieved / SUSTAINED_FP32_FMA_TFLOPS,
             "kernel": {"megakernel": "k_trace (timed together with k_primary before it and k_resolve after it)", "wavefront": "k_wf_intersect + k_wf_shade", "pixel": "k_render_mega", "bvh": "k_trace (hierarchy walk)"}[args.mode],
             "flop_per_ray": flop_per_ray, "traffic": None, "executed_achieved": None, "executed_frac": None,
             "note": "FP32 vector-ALU bound: the pair test's dot product runs on the bf16 matrix pipe (2 of ~7 SIMD cycles per sphere and wave), "
                     "verdict, accumulate, exact passes and shading on the vector ALU; HBM traffic is small"}
    if args.camera_ray_lists:
        block["frac_is"] = "NOT a brute-force figure: camera rays met only the spheres on their tile's list, so the flop count overstates the work done"
    usable = prof is not None and args.mode == "megakernel" and args.workload in ("c1", "c3", "c4") and prof.get("spheres") == n_spheres
    if usable:
        block["traffic"] = prof["hbm_bytes_per_ray"] * rays_per_launch
        block["traffic_unit"] = "HBM bytes per launch of this rank = profiled bytes per ray (rocprofv3 FETCH_SIZE x2 for coalesced reads on gfx950 + WRITE_SIZE) x rays"
        ex = prof["executed_flop_per_ray"] * rays_per_launch / (kernel_ms * 1e-3) / 1e12
        block["executed_achieved"] = ex
        block["executed_frac"] = ex / PEAK_FP32_VALU_TFLOPS
        block["valu_issue_frac"] = prof.get("valu_issue_frac")
        block["executed_is"] = ("FP32 operations the vector ALU executed per ray (profiled: FMA x 2 + ADD + MUL + transcendental wave-instructions x active lanes) x rays "
                                "/ kernel time; valu_issue_frac = share of SIMD cycles in which a vector instruction was issued")
        if prof.get("mfma_bf16_flop_per_ray"):
            block["matrix_pipe"] = {"bf16_tflops": prof["mfma_bf16_flop_per_ray"] * rays_per_launch / (kernel_ms * 1e-3) / 1e12, "busy_frac": prof.get("mfma_busy_frac"),
                                    "what": "the filter's dot products: bf16 operations the matrix pipe executed (profiled) x rays / kernel time; share of SIMD cycles the pipe was busy"}
        block["profile"] = prof["file"]
    return block


_RESULT_FD = None


def keep_stdout_for_the_result():
    """The contract is ONE JSON line on stdout.  Libraries loaded further down write there too (RCCL prints a version banner
    to stdout when a process group is created): from here on file descriptor 1 leads to stderr, and emit() writes the result line
    to the descriptor that was stdout."""
    global _RESULT_FD
    if _RESULT_FD is None:
        sys.stdout.flush()
        _RESULT_FD = os.dup(1)
        os.dup2(2, 1)


def emit(line):
    data = (json.dumps(line) + "\n").encode()
    sys.stdout.flush()
    if _RESULT_FD is None:
        os.write(1, data)
    else:
        os.write(_RESULT_FD, data)


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args, argv))
    keep_stdout_for_the_result()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's rank count and --gpus must agree")
    if args.dry_run:
        return dry_run(args, rank, world)
    # CPU baselines first, before this process touches the GPU: they start worker processes.
    cpu_lines = {}
    if world == 1 and not args.no_cpu_baseline:
        cpu_lines["cpu_baseline"] = cpu_baseline(1200, 800, 10)
        multi = cpu_baseline_multicore(1200, 800, 10, min(16, os.cpu_count() or 1))
        if multi:
            cpu_lines["cpu_baseline_multicore"] = multi
    # Full-size parity: a few rows of the 1200x800 image from the reference's own code on the keyed stream (a child process,
    # about a second of CPU), compared bit for bit with the GPU's rows further down.
    ref_rows = None
    parity_unavailable = None
    if world == 1 and not args.no_cpu_baseline and WORKLOADS[args.workload][:3] == ("final", 1200, 800):
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import ref_band
        if ref_band.exact_available():
            ref_rows = (396, 8, 4) + ref_band.keyed_band_exact(1200, 800, 4, 396, 8, args.seed)       # j0, rows, spp, image, rays
        else:
            parity_unavailable = "unavailable: oracle/_ref/ref_harness (the pinned build of the reference, made by `make -C oracle` where /root/reference exists) is not in this tree"
            print("bench.py: " + parity_unavailable, file=sys.stderr, flush=True)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the render path has no CPU fallback")
    if args.rehearse_on_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    backend = None
    use_dist = world > 1 or args.force_dist
    if use_dist:
        if world == 1:
            # a one-rank group of its own: RCCL is loaded, the gather and the reductions below run through it
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if "MASTER_PORT" not in os.environ:
                with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
                    sk.bind(("127.0.0.1", 0))
                    os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
            os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        backend = "gloo" if args.rehearse_on_one_gpu else "nccl"
        if args.rehearse_on_one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    scene_name, width, height, spp = WORKLOADS[args.workload]
    if args.spp:
        spp = args.spp
    rt = R.Renderer(local_rank, camera_ray_lists=bool(args.camera_ray_lists))
    n_spheres = setup_scene(rt, scene_name, width, height)

    band = args.band_rows
    if band <= 0:
        # one GPU: the band height is irrelevant.  Several GPUs: single rows dealt round-robin give every rank the same mix of
        # cheap and expensive rows (tools/rank_share.py: the slowest of 8 ranks is 1.5 % above the mean with 1-row bands, 4 % with
        # 4-row bands, 11 % with 8-row bands)
        band = 8 if world == 1 else 1
    mode = {"megakernel": R.MODE_MEGAKERNEL, "wavefront": R.MODE_WAVEFRONT, "pixel": R.MODE_PIXEL, "bvh": R.MODE_BVH}[args.mode]
    p = rt.params(width, height, spp, seed=args.seed, band_rows=band, band_first=rank, band_step=world, mode=mode)
    by_tiles = use_dist and args.partition == "tiles" and args.mode in ("megakernel", "bvh") and not args.rehearse_on_one_gpu
    pad = rdist.max_owned_rows(height, band, world)
    # The gathered rows are the int32 triples of main.cpp:138-140.  They are not narrowed to bytes: the reference does not clamp
    # (a NaN radiance prints as -2147483648), and 12 B per pixel is 1.4 MB per peer link on config 3 and 12.4 MB on config 4
    # (< 0.1 ms at 153 GB/s per xGMI link), so the assembled image stays identical to the single-GPU one in every case.
    out = torch.zeros((pad, width, 3), dtype=torch.int32, device=dev)
    idx = rdist.row_index(height, band, world, dev) if use_dist else None
    if by_tiles:
        # tile partition: rank r renders every world-th tile of the library's work order (rt_render_tiles_device); the image
        # tiles of every rank's buffer travel once, before the timed region, the pixels with one gather per step
        p_whole = rt.params(width, height, spp, seed=args.seed, mode=mode)
        tg = rdist.TileGather(width, height, rank, world, force_collective=use_dist)
        out = torch.zeros((tg.pad, 64, 3), dtype=torch.int32, device=dev)
        tile_ids = torch.zeros(tg.pad, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    kernel_ms, rays_per_step, gather_ms = [], [], []
    ev_a, ev_b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def step(record):
        if by_tiles:
            rt.render_tiles_device(p_whole, rank, world, d_irgb_tiles=out.data_ptr(), d_tile_ids=0 if tg.have_ids else tile_ids.data_ptr(), stream=stream)
            if not tg.have_ids:
                tg.set_ids(tile_ids)
            ev_a.record()
            img = tg.gather(out)
            ev_b.record()
            rays, ms = rt.stats()
            if record:
                kernel_ms.append(ms)
                rays_per_step.append(rays)
                ev_b.synchronize()
                gather_ms.append(ev_a.elapsed_time(ev_b))
            return img
        rt.render_device(p, d_irgb=out.data_ptr(), stream=stream)
        if use_dist:
            ev_a.record()
        if args.rehearse_on_one_gpu and world > 1:
            img = rdist.gather_rows(out.cpu(), height, band, rank, world, src_index=idx.cpu())
        else:
            img = rdist.gather_rows(out, height, band, rank, world, src_index=idx, force_collective=use_dist)
        if use_dist:
            ev_b.record()
        rays, ms = rt.stats()                 # waits for this rank's kernels (HIP events on `stream`)
        if record:
            kernel_ms.append(ms)
            rays_per_step.append(rays)
            if use_dist:
                ev_b.synchronize()
                gather_ms.append(ev_a.elapsed_time(ev_b))
        return img

    # Several ranks, row bands: the gather of step k overlaps the kernels of step k + 1.  Two output buffers alternate; the gather
    # is issued asynchronously (it starts on the collective's stream once this step's kernels are done), rank 0 puts the rows in
    # place on a side stream, and nothing in the loop waits on the host.  A buffer is rendered into again only after the gather
    # that read it -- two steps earlier -- has finished (a stream-side wait, long satisfied by then).  The steps' kernel and gather
    # times are measured afterwards, on a few steps run one after the other outside the timed region.
    overlap = use_dist and not by_tiles and not args.rehearse_on_one_gpu and not args.no_overlap
    if overlap:
        def step_overlapped(k):
            rt.render_device(p, d_irgb=pipe.buffer(k).data_ptr(), stream=stream)
            pipe.submit(k)

        try:
            # two steps outside the timed region: both buffers, the asynchronous gather and rank 0's side stream have run once.  A
            # backend that cannot do this fails here, in the same place on every rank, and the steps are then taken one by one.
            pipe = rdist.RowGatherPipeline(width, height, band, rank, world, device=dev)
            step_overlapped(0); step_overlapped(1)
            pipe.finish()
        except Exception as exc:          # noqa: BLE001 -- whatever the backend raises
            print(f"bench.py: overlapped gather unavailable ({exc!r}); stepping one by one", file=sys.stderr, flush=True)
            overlap = False

    for _ in range(args.warmup):
        step(False)
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if overlap:
        for k in range(args.steps):
            step_overlapped(k)
        img = pipe.finish()
    else:
        for _ in range(args.steps):
            img = step(True)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if overlap:
        rays_last, _ = rt.stats()
        rays_per_step.extend([rays_last] * args.steps)
        img_overlapped = img
        for _ in range(min(args.steps, 3)):        # kernel and gather times of single steps, outside the timed region
            kept = len(rays_per_step)
            step(True)
            del rays_per_step[kept:]
        img = img_overlapped

    t = torch.tensor([elapsed, float(sum(rays_per_step)), float(np.mean(kernel_ms)), float(np.max(kernel_ms))], dtype=torch.float64, device=dev)
    per_rank_kernel_ms = [float(np.mean(kernel_ms))]
    if use_dist:
        if args.rehearse_on_one_gpu:
            t = t.cpu()
        tmax = t.clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone(); dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        elapsed, total_rays, kms_mean, kms_max = tmax[0].item(), tsum[1].item(), tsum[2].item() / world, tmax[3].item()
        every = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(every, t)
        per_rank_kernel_ms = [float(e[2].item()) for e in every]
    else:
        elapsed, total_rays, kms_mean, kms_max = t[0].item(), t[1].item(), t[2].item(), t[3].item()

    if rank == 0:
        prof = profiled_counters()
        samples = width * height * spp * args.steps
        value = total_rays / elapsed / 1e6
        # dominant kernel = the render megakernel; per-launch algorithmic flops of THIS rank's launch / its mean duration
        my_rays = float(np.mean(rays_per_step))
        line = {
            "metric": "Mrays/s (1200x800 RTOW final scene)", "value": value, "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: RTOW final random-spheres scene ({n_spheres} spheres), {width}x{height}, {spp} spp, max depth 50"
                       if scene_name == "final" else (f"{args.workload}: synthetic {n_spheres}-sphere grid scene, {width}x{height}, {spp} spp, max depth 50"
                                                      if scene_name == "grid20k" else f"{args.workload}: three-sphere material scene, {width}x{height}, {spp} spp"),
                       "band_rows": band, "partition": (f"every {world}-th 8x8 tile of the work order per GPU, one RCCL gather to rank 0" if by_tiles else
                                                       f"interleaved row bands over {world} GPU(s), one RCCL gather to rank 0") if use_dist else "single GPU",
                       "seed": args.seed, "mode": args.mode,
                       "camera_rays": "per-tile sphere lists (shortcut)" if args.camera_ray_lists else "swept over all spheres like every other ray (brute force)"},
            "msamples_per_s": samples / elapsed / 1e6, "rays_per_sample": total_rays / samples,
            "kernel_ms_mean": kms_mean, "kernel_ms_max": kms_max,
            "ranks_reported_by_backend": dist.get_world_size() if use_dist else 1, "backend": (dist.get_backend() if use_dist else None),
            "kernel_ms_per_rank": per_rank_kernel_ms,
            "gather_ms_rank0": float(np.mean(gather_ms)) if gather_ms else None,
            "gather_bytes_per_step": int(out.numel() * 4 * (world - 1)) if world > 1 else 0,
            "gather_overlaps_next_step": bool(overlap),
            "roofline": roofline_block(args, n_spheres, my_rays, float(np.mean(kernel_ms)), prof),
        }
        if args.mode == "bvh":
            # the hierarchy walk skips most ray-sphere tests, so the brute-force flop count does not describe it
            line["roofline"] = {"bound": "valu", "achieved": None, "peak": PEAK_FP32_VALU_TFLOPS, "unit": "TFLOP/s", "frac": None, "traffic": None,
                                "kernel": "k_trace (hierarchy walk)",
                                "note": "optional mode beyond the brute-force sweep: per-lane tree walk, latency-bound; compare ms_per_step with the megakernel's"}
        line.update(cpu_lines)
        if ref_rows is not None:
            j0, nrows, cspp, want, _ = ref_rows
            got, _, _, _ = rt.render(rt.params(width, height, cspp, seed=args.seed, mode=mode))
            rows = got[height - (j0 + nrows):height - j0]
            same = (rows.view(np.uint32) == want.view(np.uint32)) | (np.isnan(rows) & np.isnan(want))
            line["parity_at_full_size"] = {"identical": bool(same.all()), "differing_values": int((~same).sum()),
                                           "what": f"rows {height - (j0 + nrows)}..{height - j0} of {width}x{height} at {cspp} spp, seed {args.seed}: linear RGB floats of this "
                                                   "library against the reference's own code (pinned build, no FMA contraction) on the same keyed random stream"}
        elif parity_unavailable:
            line["parity_at_full_size"] = parity_unavailable
        checksum = int(img.to(torch.int64).sum().item()) if img is not None else None
        line["image_checksum"] = checksum
        if img is not None and img.is_cuda:
            line["ppm_encode"] = ppm_encode_report(rt, img)
        if world == 1 and args.workload == "c3" and not args.spp and args.mode == "megakernel" and not args.no_config1:
            # The short launch: config 1 (1200x800 at 10 samples, the size of the reference's only published claim) on the same
            # context, outside the timed region above.  Fixed costs of a launch (fill, drain of the longest paths) weigh here.
            p1 = rt.params(width, height, WORKLOADS["c1"][3], seed=args.seed, band_rows=band, mode=mode)
            for _ in range(3):
                rt.render_device(p1, d_irgb=out.data_ptr(), stream=stream); rt.stats()
            ms1, rays1 = [], 0
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(20):
                rt.render_device(p1, d_irgb=out.data_ptr(), stream=stream)
                rays1, m = rt.stats()
                ms1.append(m)
            torch.cuda.synchronize()
            wall1 = (time.perf_counter() - t1) / 20 * 1e3
            km = float(np.mean(ms1))
            fr = rays1 * (FLOP_PER_TEST * n_spheres + FLOP_PER_RAY_EXTRA) / (km * 1e-3) / 1e12 / PEAK_FP32_VALU_TFLOPS
            line["config1"] = {"workload": f"c1: RTOW final random-spheres scene ({n_spheres} spheres), {width}x{height}, 10 spp, max depth 50",
                               "ms_per_step": wall1, "kernel_ms_mean": km, "kernel_ms_min": float(np.min(ms1)), "mrays_per_s": rays1 / (wall1 * 1e-3) / 1e6,
                               "roofline_frac": fr, "steps": 20}
        if world == 1 and args.mode == "megakernel" and not args.no_beyond and not args.camera_ray_lists:
            # Beyond brute force, outside the timed region: the same workload on a second context whose camera rays find their
            # first hit over per-tile sphere lists (rt_set_camera_ray_lists) instead of the sweep.  Same image, fewer tests.
            rt2 = R.Renderer(local_rank, camera_ray_lists=True)
            setup_scene(rt2, scene_name, width, height)
            rt2.render_device(p, d_irgb=out.data_ptr(), stream=stream); rt2.stats()
            ms2, rays2 = [], 0
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            for _ in range(min(args.steps, 5)):
                rt2.render_device(p, d_irgb=out.data_ptr(), stream=stream)
                rays2, m = rt2.stats()
                ms2.append(m)
            torch.cuda.synchronize()
            wall2 = (time.perf_counter() - t2) / len(ms2) * 1e3
            line["beyond_brute_force"] = {"what": "camera rays (38 % of all rays) against the short sphere list of their 8x8 tile instead of all spheres; every other ray swept as in the headline; same image",
                                          "ms_per_step": wall2, "kernel_ms_mean": float(np.mean(ms2)), "mrays_per_s": rays2 / (wall2 * 1e-3) / 1e6,
                                          "image_checksum_equal": int(out.to(torch.int64).sum().item()) == checksum, "rays_equal": rays2 == int(my_rays), "steps": len(ms2)}
            rt2.close()
        if line["roofline"].get("executed_frac") is not None:
            assert line["roofline"]["executed_frac"] <= 1.0, "executed FP32 rate above the peak: the profile summary under profiles/ does not belong to this kernel"
        emit(line)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
