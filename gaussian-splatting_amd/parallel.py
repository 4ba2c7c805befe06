"""Multi-GPU pieces of the path: one process per GPU, torch.distributed ("nccl" = RCCL over xGMI).

Training shards camera views over ranks with replicated Gaussians; the only exchange step is the
sum of the per-Gaussian gradients (the reference has no multi-GPU path at all: one view per
iteration, train.py:129-135).  Buckets are contiguous slices along the Gaussian axis, launched
asynchronously so that RCCL can keep several in flight on the point-to-point xGMI links; the
result equals sequential accumulation of the same views up to fp32 summation order.

Large-frame inference shards bands of tile rows: see rasterizer.render_tile_rows (no collective on
the data path, only the final gather of the bands).
"""
from __future__ import annotations

from typing import Iterable, List, Optional

import torch
import torch.distributed as dist


def shard_views(num_views: int, rank: int, world: int) -> List[int]:
    """Views rendered by `rank`: r, r + world, ...."""
    return list(range(rank, num_views, world))


def tile_row_band(tile_rows: int, rank: int, world: int):
    """Contiguous band [begin, end) of tile rows for `rank`; the first (tile_rows % world) ranks get one more."""
    q, r = divmod(tile_rows, world)
    begin = rank * q + min(rank, r)
    return begin, begin + q + (1 if rank < r else 0)


class GradReducer:
    """Bucketed all-reduce (SUM, optionally / world) of the .grad of replicated per-Gaussian parameters."""

    def __init__(self, params: Iterable[torch.Tensor], bucket_bytes: int = 64 << 20, average: bool = False,
                 group: Optional[dist.ProcessGroup] = None):
        self.params = list(params)
        self.bucket_bytes = int(bucket_bytes)
        self.average = average
        self.group = group

    def _buckets(self, g: torch.Tensor):
        g2 = g.view(g.shape[0], -1) if g.dim() > 1 else g.view(-1, 1)
        row_bytes = max(1, g2.shape[1] * g.element_size())
        rows = max(1, self.bucket_bytes // row_bytes)
        for s in range(0, g2.shape[0], rows):
            yield g2[s:s + rows]

    def all_reduce_grads(self):
        if not dist.is_initialized() or dist.get_world_size(self.group) == 1:
            return
        works = []
        for p in self.params:
            if p.grad is None:
                continue
            g = p.grad
            if not g.is_contiguous():
                g = g.contiguous()
                p.grad = g
            for b in self._buckets(g):
                works.append(dist.all_reduce(b, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        for w in works:
            w.wait()
        if self.average:
            ws = dist.get_world_size(self.group)
            for p in self.params:
                if p.grad is not None:
                    p.grad.div_(ws)


def reduce_densify_stats(xyz_gradient_accum: torch.Tensor, denom: torch.Tensor, max_radii2D: torch.Tensor,
                         group: Optional[dist.ProcessGroup] = None):
    """SUM / SUM / MAX of the densification statistics the reference keeps per Gaussian
    (scene/gaussian_model.py:710-713, train.py:185-188) across view-sharded ranks."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    dist.all_reduce(xyz_gradient_accum, op=dist.ReduceOp.SUM, group=group)
    dist.all_reduce(denom, op=dist.ReduceOp.SUM, group=group)
    dist.all_reduce(max_radii2D, op=dist.ReduceOp.MAX, group=group)


def render_band(raster_settings, rank: int, world: int, t_eps=None, **inputs):
    """This rank's share of a tile-row sharded frame: renders only its band of 16-pixel tile rows.

    Every rank holds all Gaussians; it preprocesses them all but emits / sorts / composites only the
    (tile, Gaussian) pairs of its own band, so there is no exchange on the data path.  Returns
    (image_band (3,h,W), depth_band (1,h,W), alpha_band (1,h,W), (y0, y1)); h may be 0.
    Concatenating the bands of all ranks is bit-identical to the single-GPU frame.
    """
    from .rasterizer import DEFAULT_T_EPS, GaussianRasterizer, TILE
    H = raster_settings.image_height
    tiles_y = (H + TILE - 1) // TILE
    b, e = tile_row_band(tiles_y, rank, world)
    y0, y1 = min(b * TILE, H), min(e * TILE, H)
    rast = GaussianRasterizer(raster_settings, t_eps=DEFAULT_T_EPS if t_eps is None else t_eps, tile_rows=(b, e))
    image, radii, depth, alpha = rast(**inputs)
    return image[:, y0:y1].contiguous(), depth[:, y0:y1].contiguous(), alpha[:, y0:y1].contiguous(), (y0, y1)


def gather_bands(band: torch.Tensor, H: int, dst: int = 0, group: Optional[dist.ProcessGroup] = None):
    """Collect (C,h_r,W) bands (h_r as given by tile_row_band) into the (C,H,W) frame on rank `dst`.

    Bands are padded to the tallest one so one all_gather moves them (RCCL has no ragged gather);
    returns the frame on `dst`, None elsewhere.  With no process group: returns the band itself.
    """
    from .rasterizer import TILE
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return band
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    tiles_y = (H + TILE - 1) // TILE
    spans = []
    for r in range(world):
        b, e = tile_row_band(tiles_y, r, world)
        spans.append((min(b * TILE, H), min(e * TILE, H)))
    hmax = max(y1 - y0 for y0, y1 in spans)
    C, h, W = band.shape
    assert h == spans[rank][1] - spans[rank][0], "band height does not match this rank's tile rows"
    padded = band.new_zeros((C, hmax, W))
    padded[:, :h] = band
    parts = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(parts, padded, group=group)
    if rank != dst:
        return None
    return torch.cat([parts[r][:, :spans[r][1] - spans[r][0]] for r in range(world)], dim=1)
