"""The N>1 path on CPU: two gloo ranks render interleaved bands (with the oracle standing in for the GPU
kernels) and one gather reassembles the image on rank 0; it must equal the single-process image."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import _oracle as O
from raytracing_amd import dist as rdist

W, H, SPP, SEED = 40, 27, 2, 9


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _oracle_rows(rank, world, band):
    """This rank's bands, packed in increasing y, padded to the common size."""
    scene = O.SceneArrays.from_table(O.three_sphere_table())
    cam = O.three_camera(W, H)
    pad = rdist.max_owned_rows(H, band, world)
    out = np.zeros((pad, W, 3), dtype=np.int32)
    k = 0
    for y0, y1 in rdist.owned_bands(H, band, rank, world):
        _, irgb, _ = O.render(scene, cam, W, H, SPP, O.usrc_philox(SEED), H - y1, H - y0)
        out[k:k + (y1 - y0)] = irgb
        k += y1 - y0
    return torch.from_numpy(out)


def _worker(rank, world, port, band, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    local = _oracle_rows(rank, world, band)
    img = rdist.gather_rows(local, H, band, rank, world)
    if rank == 0:
        q.put(img.numpy())
    else:
        assert img is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("band", [1, 4, 8])
def test_two_rank_gather_reassembles_image(band):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, band, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    scene = O.SceneArrays.from_table(O.three_sphere_table())
    _, want, _ = O.render(scene, O.three_camera(W, H), W, H, SPP, O.usrc_philox(SEED))
    assert np.array_equal(got, want)


def test_band_layout_covers_every_row_once():
    for height, band, world in [(800, 8, 1), (800, 8, 2), (800, 8, 8), (800, 1, 8), (800, 1, 4), (801, 1, 8), (2160, 8, 8), (27, 4, 2), (50, 16, 4), (5, 8, 8)]:
        seen = np.zeros(height, dtype=int)
        for r in range(world):
            for y0, y1 in rdist.owned_bands(height, band, r, world):
                seen[y0:y1] += 1
        assert (seen == 1).all()
        idx = rdist.row_index(height, band, world)
        assert len(set(idx.tolist())) == height
        pad = rdist.max_owned_rows(height, band, world)
        assert all(rdist.owned_row_count(height, band, r, world) <= pad for r in range(world))


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it: the script starts two ranks itself (before anything touches a
    GPU), they gather their interleaved rows over gloo and rank 0 prints the one JSON line.  --dry-run replaces the render
    kernels by a row pattern, so this runs without a GPU."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run", "--workload", "c1"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["ranks_reported_by_backend"] == 2 and line["assembled_image_ok"] is True
    # a rank count that disagrees with --gpus is refused
    env2 = dict(env, WORLD_SIZE="3", RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run"], capture_output=True, text=True, timeout=120, env=env2, cwd=root)
    assert r.returncode != 0 and "must agree" in (r.stderr + r.stdout)


# ---------------------------------------------------------------------------- tile partition
def _tile_share_of(img, order, rank, world):
    """What rt_render_tiles_device hands rank `rank`: every world-th tile of `order` from position rank on, tile-major, padded."""
    h, w = img.shape[0], img.shape[1]
    tx, ty = rdist.tile_grid(w, h)
    padded = np.zeros((ty * 8, tx * 8, 3), dtype=np.int32)
    padded[:h, :w] = img
    tiles = padded.reshape(ty, 8, tx, 8, 3).transpose(0, 2, 1, 3, 4).reshape(ty * tx, 64, 3)
    mine = order[rank::world]
    pad = rdist.max_owned_tiles(w, h, world)
    out = np.zeros((pad, 64, 3), dtype=np.int32)
    ids = np.zeros(pad, dtype=np.int32)
    out[:len(mine)] = tiles[mine]
    ids[:len(mine)] = mine
    return torch.from_numpy(out), torch.from_numpy(ids)


def _tile_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    scene = O.SceneArrays.from_table(O.three_sphere_table())
    _, whole, _ = O.render(scene, O.three_camera(W, H), W, H, SPP, O.usrc_philox(SEED))
    tx, ty = rdist.tile_grid(W, H)
    order = np.random.default_rng(5).permutation(tx * ty)           # stands in for the library's class-sorted work order
    g = rdist.TileGather(W, H, rank, world)
    for step in range(2):                                            # the ids travel once, the pixels every step
        tiles, ids = _tile_share_of(whole + step, order, rank, world)
        if not g.have_ids:
            g.set_ids(ids)
        img = g.gather(tiles)
        if rank == 0:
            q.put(img.numpy())
        else:
            assert img is None
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_tile_gather_reassembles_image():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_tile_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120), q.get(timeout=120)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    scene = O.SceneArrays.from_table(O.three_sphere_table())
    _, want, _ = O.render(scene, O.three_camera(W, H), W, H, SPP, O.usrc_philox(SEED))
    assert np.array_equal(got[0], want) and np.array_equal(got[1], want + 1)


def test_tile_layout_covers_every_tile_once():
    for width, height, world in [(1200, 800, 1), (1200, 800, 8), (3840, 2160, 8), (40, 27, 2), (13, 5, 4), (8, 8, 3)]:
        tx, ty = rdist.tile_grid(width, height)
        counts = [rdist.owned_tile_count(width, height, r, world) for r in range(world)]
        assert sum(counts) == tx * ty and max(counts) == rdist.max_owned_tiles(width, height, world) and max(counts) - min(counts) <= 1
        order = np.random.default_rng(width).permutation(tx * ty)
        img = np.arange(height * width * 3, dtype=np.int32).reshape(height, width, 3)
        parts = [_tile_share_of(img, order, r, world) for r in range(world)]
        tiles = torch.cat([t[:n] for (t, _), n in zip(parts, counts)])
        ids = torch.cat([i[:n] for (_, i), n in zip(parts, counts)])
        assert np.array_equal(rdist.assemble_tiles(tiles, ids, width, height).numpy(), img)


def _pipeline_worker(rank, world, port, band, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pipe = rdist.RowGatherPipeline(W, H, band, rank, world)
    steps, bad = 5, []

    def want(k):
        return ((torch.arange(H, dtype=torch.int32)[:, None, None] * 7 + k * 1000 + torch.arange(3, dtype=torch.int32)[None, None, :]) % 65521).expand(H, W, 3)

    for k in range(steps):
        buf = pipe.buffer(k)               # (free again: the gather of step k - 2 has read it)
        buf.zero_()
        j = 0
        for y0, y1 in rdist.owned_bands(H, band, rank, world):
            buf[j:j + (y1 - y0)] = want(k)[y0:y1]
            j += y1 - y0
        pipe.submit(k)
        if k >= 1:
            img = pipe.image(k - 1)        # the step before, while this step's gather may still be under way
            if rank == 0 and not torch.equal(img, want(k - 1)):
                bad.append(k - 1)
    last = pipe.finish()
    if rank == 0:
        if not torch.equal(last, want(steps - 1)):
            bad.append(steps - 1)
        q.put(bad)
    else:
        assert last is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("band", [1, 8])
def test_two_rank_pipeline_overlaps_gather_and_next_step(band):
    """bench.py's several-GPU loop: two output buffers, the gather of step k issued asynchronously while step k + 1 fills the
    other buffer; every step's assembled image on rank 0 is that step's."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pipeline_worker, args=(r, world, port, band, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    bad = q.get(timeout=120)
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    assert bad == []
