"""Row-band partition of one image over ranks and the single gather that reassembles it.

Every (pixel, sample) is independent and the scene and camera are replicated, so the only
exchange step of the path is the gather of finished rows to rank 0.  Bands are dealt round-robin
(band b -> rank b mod world) so that cheap sky rows and expensive sphere rows spread evenly.
Works on CUDA tensors over RCCL (backend "nccl") and on CPU tensors over gloo.
"""
import torch
import torch.distributed as dist


def owned_bands(height, band_rows, rank, world):
    """[(y0, y1), ...] image-row ranges (from the top) that `rank` renders, in increasing y."""
    nb = (height + band_rows - 1) // band_rows
    return [(b * band_rows, min((b + 1) * band_rows, height)) for b in range(rank, nb, world)]


def owned_row_count(height, band_rows, rank, world):
    return sum(y1 - y0 for y0, y1 in owned_bands(height, band_rows, rank, world))


def max_owned_rows(height, band_rows, world):
    return max(owned_row_count(height, band_rows, r, world) for r in range(world))


def row_index(height, band_rows, world, device=None):
    """For the rank-major concatenation of padded per-rank buffers: index tensor `src` such that
    image[y] = stacked[src[y]]."""
    pad = max_owned_rows(height, band_rows, world)
    src = torch.empty(height, dtype=torch.long)
    for r in range(world):
        k = 0
        for y0, y1 in owned_bands(height, band_rows, r, world):
            src[y0:y1] = torch.arange(r * pad + k, r * pad + k + (y1 - y0))
            k += y1 - y0
    return src.to(device) if device is not None else src


def gather_rows(local, height, band_rows, rank, world, group=None, src_index=None, force_collective=False):
    """local: [max_owned_rows, width, C] tensor whose first owned_row_count rows are valid.
    Returns the assembled [height, width, C] image on rank 0 and None elsewhere.  A single rank returns its rows as they
    are unless force_collective asks for the gather to run anyway (a one-rank process group: the backend is loaded and the
    collective executes, which is how a one-GPU box rehearses the RCCL path)."""
    if world == 1 and not force_collective:
        return local[:height]
    pad = max_owned_rows(height, band_rows, world)
    assert local.shape[0] == pad, (local.shape, pad)
    if rank == 0:
        parts = [torch.empty_like(local) for _ in range(world)]
        dist.gather(local, parts, dst=0, group=group)
        stacked = torch.cat(parts, dim=0)
        idx = src_index if src_index is not None else row_index(height, band_rows, world, local.device)
        return stacked.index_select(0, idx)
    dist.gather(local, None, dst=0, group=group)
    return None


def render_image(rt, width, height, spp, seed=0, band_rows=8, rank=0, world=1, device=None, out=None, src_index=None,
                 group=None, mode=0):
    """Render this rank's bands with `rt` (a Renderer on this rank's GPU) into a device tensor and gather.
    Returns (image on rank 0 | None, local tensor).  The kernels run on torch's current stream, so the
    collective that follows is ordered after them."""
    p = rt.params(width, height, spp, seed=seed, band_rows=band_rows, band_first=rank, band_step=world, mode=mode)
    pad = max_owned_rows(height, band_rows, world)
    if out is None:
        out = torch.zeros((pad, width, 3), dtype=torch.int32, device=device)
    rt.render_device(p, d_irgb=out.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
    return gather_rows(out, height, band_rows, rank, world, group=group, src_index=src_index), out
