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
