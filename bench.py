#!/usr/bin/env python3
"""Throughput of the render loop on MI355X: Mrays/s on the 1200x800 RTOW final scene.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c3|c1|c2|c5]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = one render of the whole image: each rank renders its interleaved row bands (scene and
camera replicated, resident in HBM before the timed region) and, for N > 1, one RCCL gather
assembles the rows on rank 0.  Total work is fixed as N grows ("strong" scaling).  Rank 0 prints
one JSON line.
"""
import argparse
import ctypes as C
import json
import os
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
# HBM bytes per ray of the render kernels from rocprofv3 FETCH_SIZE (x2, the gfx950 correction for coalesced reads) + WRITE_SIZE
# (profiles/r01_pmc_traffic.txt); a profiled constant, not measured inside this script
HBM_BYTES_PER_RAY_PMC = 44.9

WORKLOADS = {
    # name: (scene, width, height, spp)
    "c1": ("final", 1200, 800, 10),
    "c2": ("three", 400, 225, 100),
    "c3": ("final", 1200, 800, 500),
    "c4": ("final", 3840, 2160, 1024),
    "c5": ("grid20k", 1920, 1080, 256),
}


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--spp", type=int, default=0, help="override samples per pixel")
    ap.add_argument("--band-rows", type=int, default=0, help="rows per band (0 = 8 on one GPU, 1 on several)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--mode", default="megakernel", choices=["megakernel", "wavefront", "pixel", "bvh"])
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="debug: all ranks share GPU 0 and gather over gloo (RCCL refuses two ranks on one device)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
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
    if world == 1 and not args.no_cpu_baseline and WORKLOADS[args.workload][:3] == ("final", 1200, 800):
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import ref_band
        if ref_band.exact_available():
            ref_rows = (396, 8, 4) + ref_band.keyed_band_exact(1200, 800, 4, 396, 8, args.seed)       # j0, rows, spp, image, rays
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the render path has no CPU fallback")
    if args.rehearse_on_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if args.rehearse_on_one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    scene_name, width, height, spp = WORKLOADS[args.workload]
    if args.spp:
        spp = args.spp
    rt = R.Renderer(local_rank)
    if scene_name == "final":
        table = R.final_scene_table()
        rt.upload_scene(table)
        rt.set_camera(R.final_camera(width, height))
    elif scene_name == "grid20k":
        table = R.scene_grid(42, 75, 19996)
        rt.upload_scene(table)
        rt.set_camera(R.final_camera(width, height))
    else:
        table = R.three_sphere_table()
        rt.upload_scene(table)
        rt.set_camera(R.three_camera(width, height))
    n_spheres = len(table)

    band = args.band_rows
    if band <= 0:
        # one GPU: the band height is irrelevant.  Several GPUs: single rows dealt round-robin give every rank the same mix of
        # cheap and expensive rows (tools/rank_share.py: the slowest of 8 ranks is 1.5 % above the mean with 1-row bands, 4 % with
        # 4-row bands, 11 % with 8-row bands)
        band = 8 if world == 1 else 1
    mode = {"megakernel": R.MODE_MEGAKERNEL, "wavefront": R.MODE_WAVEFRONT, "pixel": R.MODE_PIXEL, "bvh": R.MODE_BVH}[args.mode]
    p = rt.params(width, height, spp, seed=args.seed, band_rows=band, band_first=rank, band_step=world, mode=mode)
    pad = rdist.max_owned_rows(height, band, world)
    out = torch.zeros((pad, width, 3), dtype=torch.int32, device=dev)
    idx = rdist.row_index(height, band, world, dev) if world > 1 else None
    stream = torch.cuda.current_stream().cuda_stream

    kernel_ms, rays_per_step = [], []

    def step(record):
        rt.render_device(p, d_irgb=out.data_ptr(), stream=stream)
        if args.rehearse_on_one_gpu and world > 1:
            img = rdist.gather_rows(out.cpu(), height, band, rank, world, src_index=idx.cpu())
        else:
            img = rdist.gather_rows(out, height, band, rank, world, src_index=idx)
        rays, ms = rt.stats()                 # waits for this rank's kernels (HIP events on `stream`)
        if record:
            kernel_ms.append(ms)
            rays_per_step.append(rays)
        return img

    for _ in range(args.warmup):
        step(False)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        img = step(True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0

    t = torch.tensor([elapsed, float(sum(rays_per_step)), float(np.mean(kernel_ms)), float(np.max(kernel_ms))], dtype=torch.float64, device=dev)
    if world > 1:
        if args.rehearse_on_one_gpu:
            t = t.cpu()
        tmax = t.clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone(); dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        elapsed, total_rays, kms_mean, kms_max = tmax[0].item(), tsum[1].item(), tsum[2].item() / world, tmax[3].item()
    else:
        elapsed, total_rays, kms_mean, kms_max = t[0].item(), t[1].item(), t[2].item(), t[3].item()

    if rank == 0:
        samples = width * height * spp * args.steps
        value = total_rays / elapsed / 1e6
        flop_per_ray = FLOP_PER_TEST * n_spheres + FLOP_PER_RAY_EXTRA
        # dominant kernel = the render megakernel; per-launch algorithmic flops of THIS rank's launch / its mean duration
        my_rays = float(np.mean(rays_per_step))
        achieved = my_rays * flop_per_ray / (float(np.mean(kernel_ms)) * 1e-3) / 1e12
        line = {
            "metric": "Mrays/s (1200x800 RTOW final scene)", "value": value, "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: RTOW final random-spheres scene ({n_spheres} spheres), {width}x{height}, {spp} spp, max depth 50"
                       if scene_name == "final" else (f"{args.workload}: synthetic {n_spheres}-sphere grid scene, {width}x{height}, {spp} spp, max depth 50"
                                                      if scene_name == "grid20k" else f"{args.workload}: three-sphere material scene, {width}x{height}, {spp} spp"),
                       "band_rows": band, "partition": f"interleaved row bands over {world} GPU(s), one RCCL gather to rank 0" if world > 1 else "single GPU",
                       "seed": args.seed, "mode": args.mode},
            "msamples_per_s": samples / elapsed / 1e6, "rays_per_sample": total_rays / samples,
            "kernel_ms_mean": kms_mean, "kernel_ms_max": kms_max,
            "roofline": {"bound": "valu", "achieved": achieved, "peak": PEAK_FP32_VALU_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_FP32_VALU_TFLOPS,
                         "peak_sustained_measured": SUSTAINED_FP32_FMA_TFLOPS, "frac_of_sustained": achieved / SUSTAINED_FP32_FMA_TFLOPS,
                         "traffic": (HBM_BYTES_PER_RAY_PMC * my_rays) if args.mode == "megakernel" else None, "traffic_unit": "bytes per step of this rank (PMC-derived constant per ray)",
                         "kernel": {"megakernel": "k_trace", "wavefront": "k_wf_intersect + k_wf_shade", "pixel": "k_render_mega", "bvh": "k_trace (hierarchy walk)"}[args.mode], "flop_per_ray": flop_per_ray,
                         "note": "FP32 vector-ALU bound (HBM traffic and MFMA are not on this path); algorithmic flops = rays x (18 x spheres + 150)"},
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
            band = got[height - (j0 + nrows):height - j0]
            same = (band.view(np.uint32) == want.view(np.uint32)) | (np.isnan(band) & np.isnan(want))
            line["parity_at_full_size"] = {"identical": bool(same.all()), "differing_values": int((~same).sum()),
                                           "what": f"rows {height - (j0 + nrows)}..{height - j0} of {width}x{height} at {cspp} spp, seed {args.seed}: linear RGB floats of this "
                                                   "library against the reference's own code (pinned build, no FMA contraction) on the same keyed random stream"}
        checksum = int(img.to(torch.int64).sum().item()) if img is not None else None
        line["image_checksum"] = checksum
        if img is not None and img.is_cuda:
            line["ppm_encode"] = ppm_encode_report(rt, img)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
