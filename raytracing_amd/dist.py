"""Partition of one image over ranks and the single gather that reassembles it.

Every (pixel, sample) is independent and the scene and camera are replicated, so the only
exchange step of the path is the gather of finished pixels to rank 0.  Two partitions:

* by tiles (default of bench.py for N > 1): the library orders the 8x8 tiles of the image by how long their
  paths can get and rank r renders every world-th tile of that order from position r on
  (rt_render_tiles_device), so every rank holds the same mix of expensive and cheap tiles and tiles
  stay compact; rank 0 gathers the tile-major buffers and puts the tiles in place by index;
* by interleaved row bands (band b -> rank b mod world), the partition rt_params itself describes.

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


class RowGatherPipeline:
    """Row bands, several steps in a row: the gather of step k runs beside the rendering of step k + 1.

    Two output buffers alternate.  `buffer(k)` hands out the buffer step k renders into -- first making the caller's stream wait
    (CUDA) or the host wait (CPU tensors) for the gather that read it two steps earlier; `submit(k)` issues the gather of that buffer
    asynchronously (on CUDA it starts on the collective's stream once the kernels queued so far on the current stream are done) and,
    on rank 0, puts the gathered rows in place -- on a side stream for CUDA tensors, so that the current stream goes straight on to
    the next step; `finish()` waits for everything and returns the last step's image on rank 0 (None elsewhere).  Works over RCCL
    (backend "nccl") on CUDA tensors and over gloo on CPU tensors."""

    def __init__(self, width, height, band_rows, rank, world, device=None, channels=3, dtype=torch.int32, group=None):
        self.width, self.height, self.band_rows, self.rank, self.world, self.group = width, height, band_rows, rank, world, group
        self.pad = max_owned_rows(height, band_rows, world)
        dev = torch.device("cpu") if device is None else torch.device(device)
        self.cuda = dev.type == "cuda"
        self.outs = [torch.zeros((self.pad, width, channels), dtype=dtype, device=dev) for _ in range(2)]
        root = rank == 0
        self.recv = [torch.empty((world, self.pad, width, channels), dtype=dtype, device=dev) if root else None for _ in range(2)]
        self.imgs = [torch.empty((height, width, channels), dtype=dtype, device=dev) if root else None for _ in range(2)]
        self.idx = row_index(height, band_rows, world, dev) if root else None
        self.pending = [None, None]
        self.unplaced = [False, False]          # CPU tensors, rank 0: gathered rows of that buffer not yet put in place
        self.last = None
        if self.cuda:
            self.side = torch.cuda.Stream(device=dev)
            self.placed = [torch.cuda.Event(), torch.cuda.Event()]

    def _place(self, b):
        torch.index_select(self.recv[b].view(self.world * self.pad, self.width, -1), 0, self.idx, out=self.imgs[b])

    def _settle(self, b):
        """The gather of buffer b is over as far as the caller's stream (CUDA) or the host (CPU) is concerned, and rank 0 has read recv[b]."""
        if self.pending[b] is None:
            return
        self.pending[b].wait()
        if self.rank == 0:
            if self.cuda:
                torch.cuda.current_stream().wait_event(self.placed[b])
            elif self.unplaced[b]:
                self._place(b)
                self.unplaced[b] = False
        self.pending[b] = None

    def buffer(self, k):
        b = k & 1
        self._settle(b)
        return self.outs[b]

    def submit(self, k):
        b = k & 1
        self.pending[b] = dist.gather(self.outs[b], list(self.recv[b].unbind(0)) if self.rank == 0 else None, dst=0, group=self.group, async_op=True)
        if self.rank == 0:
            if self.cuda:
                with torch.cuda.stream(self.side):
                    self.pending[b].wait()
                    self._place(b)
                    self.placed[b].record(self.side)
            else:
                self.unplaced[b] = True
        self.last = b

    def image(self, k):
        """Rank 0: the assembled image of step k (k must be one of the last two steps submitted); waits for it."""
        b = k & 1
        self._settle(b)
        if self.cuda:
            self.side.synchronize()
        return self.imgs[b]

    def finish(self):
        for b in (0, 1):
            self._settle(b)
        if self.cuda:
            self.side.synchronize()
        return self.imgs[self.last] if self.rank == 0 and self.last is not None else None


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


# ---------------------------------------------------------------------------- tile partition
def tile_grid(width, height):
    """(tiles_x, tiles_y) of 8x8 tiles, edge tiles padded."""
    return (width + 7) // 8, (height + 7) // 8


def owned_tile_count(width, height, rank, world):
    tx, ty = tile_grid(width, height)
    n = tx * ty
    return (n - rank + world - 1) // world if n > rank else 0


def max_owned_tiles(width, height, world):
    return owned_tile_count(width, height, 0, world)


def assemble_tiles(tiles, ids, width, height):
    """tiles: [n, 64, C] tile-major pixels (pixel l of a tile at row l // 8, column l % 8), ids: [n] image tile of each
    (row-major over the tile grid); every tile of the image exactly once.  Returns the [height, width, C] image."""
    tx, ty = tile_grid(width, height)
    assert tiles.shape[0] == tx * ty and ids.shape[0] == tx * ty, (tiles.shape, ids.shape, tx * ty)
    full = torch.empty_like(tiles)
    full[ids.long()] = tiles
    c = tiles.shape[-1]
    return full.view(ty, tx, 8, 8, c).permute(0, 2, 1, 3, 4).reshape(ty * 8, tx * 8, c)[:height, :width]


class TileGather:
    """The gather of a tile partition.  Which image tile sits where in every rank's buffer depends on scene, camera and image size
    only, so the ids travel once (`set_ids`, one gather of 4 bytes per tile) and every step after that is ONE gather of pixels."""

    def __init__(self, width, height, rank, world, group=None, force_collective=False):
        self.width, self.height, self.rank, self.world, self.group = width, height, rank, world, group
        self.collective = world > 1 or force_collective
        self.pad = max_owned_tiles(width, height, world)
        self.counts = [owned_tile_count(width, height, r, world) for r in range(world)]
        self.ids = None            # rank 0: [tiles] image tile of every gathered tile, rank-major
        self.have_ids = False      # every rank: set_ids() has run (all ranks take part in its gather)

    def _gather(self, local):
        if not self.collective:
            return [local]
        if self.rank == 0:
            parts = [torch.empty_like(local) for _ in range(self.world)]
            dist.gather(local, parts, dst=0, group=self.group)
            return parts
        dist.gather(local, None, dst=0, group=self.group)
        return None

    def set_ids(self, local_ids):
        """local_ids: [pad] int32, the first owned_tile_count entries valid."""
        assert local_ids.shape[0] == self.pad, (local_ids.shape, self.pad)
        parts = self._gather(local_ids)
        self.have_ids = True
        if self.rank == 0:
            self.ids = torch.cat([p[:n] for p, n in zip(parts, self.counts)])

    def gather(self, local_tiles):
        """local_tiles: [pad, 64, C].  Returns the assembled image on rank 0, None elsewhere."""
        assert local_tiles.shape[0] == self.pad, (local_tiles.shape, self.pad)
        parts = self._gather(local_tiles)
        if self.rank != 0:
            return None
        assert self.ids is not None, "set_ids() first"
        return assemble_tiles(torch.cat([p[:n] for p, n in zip(parts, self.counts)]), self.ids, self.width, self.height)


def render_image_tiles(rt, width, height, spp, seed=0, rank=0, world=1, device=None, gatherer=None, out=None, mode=0, force_collective=False):
    """Render this rank's tile share with `rt` into a device tensor and gather.  Returns (image on rank 0 | None, gatherer, local tiles)."""
    p = rt.params(width, height, spp, seed=seed, mode=mode)
    if gatherer is None:
        gatherer = TileGather(width, height, rank, world, force_collective=force_collective)
    if out is None:
        out = torch.zeros((gatherer.pad, 64, 3), dtype=torch.int32, device=device)
    need_ids = not gatherer.have_ids
    ids = torch.zeros(gatherer.pad, dtype=torch.int32, device=device) if need_ids else None
    rt.render_tiles_device(p, rank, world, d_irgb_tiles=out.data_ptr(), d_tile_ids=ids.data_ptr() if need_ids else 0,
                           stream=torch.cuda.current_stream().cuda_stream)
    if need_ids:
        gatherer.set_ids(ids)
    return gatherer.gather(out), gatherer, out
