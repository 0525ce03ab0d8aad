"""RCCL on the GPU box: a one-rank process group in a fresh child process.  No scaling is measured here (the box has one
GPU); the test proves that the library loads, that the gather of device-resident rows and the reductions bench.py uses run
through backend "nccl" (= RCCL on ROCm) on a device tensor, and that the group tears down cleanly."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import json, os, sys
import torch
import torch.distributed as dist
sys.path.insert(0, os.environ["RT_ROOT"])
import raytracing_amd as R
from raytracing_amd import dist as rdist

torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
W, H, S = 96, 64, 4
rt = R.Renderer(0)
rt.upload_scene(R.final_scene_table())
rt.set_camera(R.final_camera(W, H))
img, local = rdist.render_image(rt, W, H, S, seed=3, band_rows=1, rank=0, world=1, device=dev)
rays, ms = rt.stats()
pad = rdist.max_owned_rows(H, 1, 1)
gathered = rdist.gather_rows(local, H, 1, 0, 1, force_collective=True)      # dist.gather on the device tensor
t = torch.tensor([float(rays), ms], dtype=torch.float64, device=dev)
tiled, tg, _ = rdist.render_image_tiles(rt, W, H, S, seed=3, rank=0, world=1, device=dev, force_collective=True)      # the tile partition's two gathers
tmax = t.clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
tsum = t.clone(); dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
every = [torch.zeros_like(t)]
dist.all_gather(every, t)
dist.barrier()
torch.cuda.synchronize()
_, want, want_rays, _ = rt.render(rt.params(W, H, S, seed=3))
out = {"backend": dist.get_backend(), "world": dist.get_world_size(), "gathered_is_cuda": bool(gathered.is_cuda),
       "gather_equals_local": bool(torch.equal(gathered, local[:H])), "image_equals_host_form": bool((gathered.cpu().numpy() == want).all()), "tile_image_equals_host_form": bool((tiled.cpu().numpy() == want).all()),
       "rays": int(rays), "rays_host_form": int(want_rays), "allreduce_ok": bool(torch.equal(tmax, t) and torch.equal(tsum, t) and torch.equal(every[0], t))}
dist.destroy_process_group()
print("RESULT " + json.dumps(out), flush=True)
"""


def _free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_one_rank_nccl_group_gathers_device_rows():
    env = dict(os.environ)
    env.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(_free_port()), "RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0",
                "RT_ROOT": ROOT})
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=300)
    assert proc.returncode == 0, proc.stderr[-2000:]
    line = [l for l in proc.stdout.splitlines() if l.startswith("RESULT ")]
    assert line, proc.stdout[-2000:]
    out = json.loads(line[-1][7:])
    assert out["backend"] == "nccl" and out["world"] == 1
    assert out["gathered_is_cuda"] and out["gather_equals_local"] and out["image_equals_host_form"] and out["tile_image_equals_host_form"]
    assert out["rays"] == out["rays_host_form"] and out["allreduce_ok"]


def test_bench_line_through_a_one_rank_nccl_group():
    """bench.py --force-dist: the N = 1 line with the process group in place (a short run of the small workload)."""
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    lines = {}
    for name, extra in (("overlapped", []), ("step by step", ["--no-overlap"])):
        # (five steps: the two output buffers of the overlapped form are each used more than once)
        proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--force-dist", "--workload", "c1", "--steps", "5", "--warmup", "1",
                               "--no-cpu-baseline", "--no-beyond"] + extra, env=env, capture_output=True, text=True, timeout=300)
        assert proc.returncode == 0, proc.stderr[-2000:]
        line = json.loads(proc.stdout.strip().splitlines()[-1])
        assert line["backend"] == "nccl" and line["ranks_reported_by_backend"] == 1 and line["n_gpus"] == 1
        assert line["gather_ms_rank0"] is not None and line["value"] > 0
        lines[name] = line
    # the gather of step k running beside the kernels of step k + 1 changes neither the image nor the ray count
    assert lines["overlapped"]["gather_overlaps_next_step"] and not lines["step by step"]["gather_overlaps_next_step"]
    assert lines["overlapped"]["image_checksum"] == lines["step by step"]["image_checksum"]
    assert lines["overlapped"]["rays_per_sample"] == lines["step by step"]["rays_per_sample"]
