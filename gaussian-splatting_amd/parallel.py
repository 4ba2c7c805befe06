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

import os
from typing import Iterable, List, Optional

import torch
import torch.distributed as dist

from .profiling import stage as _stage

_DEBUG = bool(os.environ.get("GS_DEBUG_REDUCE"))    # prints every bucket before its all-reduce


def shard_views(num_views: int, rank: int, world: int) -> List[int]:
    """Views rendered by `rank`: r, r + world, ...."""
    return list(range(rank, num_views, world))


def tile_row_band(tile_rows: int, rank: int, world: int):
    """Contiguous band [begin, end) of tile rows for `rank`; the first (tile_rows % world) ranks get one more."""
    q, r = divmod(tile_rows, world)
    begin = rank * q + min(rank, r)
    return begin, begin + q + (1 if rank < r else 0)


def _all_reduce_sum(tensors, group, async_op: bool):
    """SUM all-reduce of several tensors, in place.  RCCL: one grouped launch (a single ncclGroup) on RCCL's own
    stream, ordered after what the current stream holds at this point; async_op returns a handle for _wait().
    gloo (tests and rehearsals only: several ranks on one GPU, or CPU tensors): device tensors are staged through
    the host by hand, one at a time -- gloo's own device path (pinned staging allocated per call, a worker
    thread synchronising side streams) stalls for minutes with two busy processes on one GPU."""
    tensors = [t for t in tensors if t is not None and t.numel() > 0]
    if not tensors:
        return None
    if dist.get_backend(group) == "gloo":
        for t in tensors:
            if t.is_cuda:
                h = t.detach().cpu()
                dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
                t.copy_(h)
            else:
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        return None
    if len(tensors) == 1:
        w = dist.all_reduce(tensors[0], op=dist.ReduceOp.SUM, group=group, async_op=async_op)
        return w if async_op else None
    cm_factory = getattr(dist.distributed_c10d, "_coalescing_manager", None)
    if cm_factory is not None:
        with cm_factory(group=group, device=tensors[0].device, async_ops=async_op) as cm:
            for t in tensors:
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        return cm if async_op else None
    works = [dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group, async_op=async_op) for t in tensors]
    return works if async_op else None


class GradReducer:
    """Bucketed all-reduce (SUM, optionally / world) of the .grad of replicated per-Gaussian parameters, after the
    backward has finished (ViewShardedRasterizer below overlaps the same exchange with the backward)."""

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
                if _DEBUG:
                    print(f"[GradReducer] rank {dist.get_rank(self.group)}: {tuple(b.shape)} {b.dtype} {b.device} "
                          f"contiguous={b.is_contiguous()} offset={b.storage_offset()}", flush=True)
                works.append(_all_reduce_sum([b], self.group, async_op=True))   # several buckets in flight on RCCL's stream
        for w in works:
            _wait(w)
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


# ---------------------------------------------------------------------------------------------------
# Gaussian-sharded training step: exchange splat records, not gradients
# ---------------------------------------------------------------------------------------------------
def _all_to_all_rows(out: torch.Tensor, inp: torch.Tensor, out_rows, in_rows, group, async_op: bool = False):
    """all_to_all_single along dim 0 with per-peer row counts.  RCCL (and gloo on CPU tensors) take it
    directly; gloo has no device all-to-all, so device tensors are staged through the host there (that
    combination only exists in tests: two ranks on one GPU).  async_op: returns a handle for _wait()
    (None when the transfer already completed)."""
    if dist.get_backend(group) == "gloo" and inp.is_cuda:
        o = torch.empty(out.shape, dtype=out.dtype)
        dist.all_to_all_single(o, inp.cpu(), list(out_rows), list(in_rows), group=group)
        out.copy_(o)
        return None
    w = dist.all_to_all_single(out, inp, list(out_rows), list(in_rows), group=group, async_op=async_op)
    return w if async_op else None


def _wait(work):
    """Make the current stream wait for an asynchronous collective (no host block with RCCL)."""
    if work is None:
        return
    for w in (work if isinstance(work, (list, tuple)) else [work]):
        w.wait()


class _RasterizeSharded(torch.autograd.Function):
    """One training step over `world` views with the Gaussians sharded over ranks (see SplatExchange)."""

    @staticmethod
    def forward(ctx, means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp,
                views, t_eps, counts, group, need_grad=True):
        from . import rasterizer as R
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        dev = means3D.device
        n = means3D.shape[0]
        total = int(sum(counts))
        mine = views[rank]
        i32 = dict(dtype=torch.int32, device=dev)
        with torch.cuda.device(dev):
            # 1. project my shard into every rank's view, straight into the send buffers
            splat_s = torch.empty((world, max(n, 1), 12), dtype=torch.float32, device=dev)[:, :n]
            dkeys_s = torch.empty((world, max(n, 1)), **i32)[:, :n]
            rects_s = torch.empty((world, max(n, 1), 2), **i32)[:, :n]
            ids_scratch = torch.empty((max(n, 1),), **i32)[:n]
            # the colour side table of my shard in every view stays here, for the per-view chain rule of the backward
            jac = None
            if need_grad and R.USE_SH_JAC and shs is not None and n > 0:
                jac = torch.empty((world, n, R.sh_jac_row_floats()), dtype=torch.float32, device=dev)
            # (one launch for all views: gs_preprocess_fwd_views reads the shard's coefficients once; every view writes the
            # same ids into the scratch row, which nothing reads)
            pre = R.preprocess_views(views, means3D, opacities, scales, rotations, cov3D_precomp, shs, colors_precomp,
                                     outs=[(splat_s[v], dkeys_s[v], ids_scratch, rects_s[v]) for v in range(world)],
                                     sh_jacs=None if jac is None else [jac[v] for v in range(world)])
            radii = torch.stack([p[1] for p in pre]) if world else torch.empty((0, n), **i32)
            del pre
            # 2. records, depth keys and boxes of ALL Gaussians for MY view, in global index order
            splat = torch.empty((max(total, 1), 12), dtype=torch.float32, device=dev)[:total]
            dkeys = torch.empty((2, max(total, 1)), **i32)
            ids = torch.empty((2, max(total, 1)), **i32)
            rects = torch.empty((max(total, 1), 2), **i32)[:total]
            same = [n] * world
            # depth keys first (the depth sort needs only them), then the boxes (needed from the scan on), then
            # the 48-byte records (needed by the compositing kernel): the big transfer runs on RCCL's stream
            # while this stream sorts and bins
            with _stage("exchange_fwd"):
                w_keys = _all_to_all_rows(dkeys[0][:total], dkeys_s.reshape(world * n), counts, same, group, True)
                w_rects = _all_to_all_rows(rects, rects_s.reshape(world * n, 2), counts, same, group, True)
                w_splat = _all_to_all_rows(splat, splat_s.reshape(world * n, 12), counts, same, group, True)
            torch.arange(total, out=ids[0][:total], **i32)
            _wait(w_keys)
            # 3. bin and composite my view
            state = R.bin_splats(mine, splat, None, None, dkeys, ids, rects, before_scan=lambda: _wait(w_rects))
            with _stage("exchange_fwd_wait"):
                _wait(w_splat)
            band = R._band(mine, None)
            image, depth, alpha, aux, rows = R.render_forward(mine, splat, state.ranges, state.point_list, need_grad,
                                                              t_eps, band)
        ctx.rows = rows
        ctx.set_materialize_grads(False)
        ctx.views, ctx.t_eps, ctx.counts, ctx.group, ctx.band = views, float(t_eps), list(counts), group, band
        ctx.has = tuple(t is not None for t in (shs, colors_precomp, scales, rotations, cov3D_precomp))
        ctx.means2D_shape = None if means2D is None else tuple(means2D.shape)
        saved = [means3D, opacities, splat, radii, state.point_list, state.ranges, depth, alpha]
        saved += [t for t in (shs, colors_precomp, scales, rotations, cov3D_precomp) if t is not None]
        if aux is not None:
            saved.append(aux)
        ctx.has_aux = aux is not None
        ctx.has_jac = jac is not None        # saved like the rest: freed with the graph (rasterizer._RasterizeGaussians)
        if ctx.has_jac:
            saved.append(jac)
        ctx.save_for_backward(*saved)
        ctx.mark_non_differentiable(radii)
        return image, radii, depth, alpha

    @staticmethod
    def backward(ctx, g_image, g_radii, g_depth, g_alpha):
        from . import rasterizer as R
        saved = list(ctx.saved_tensors)
        means3D, opacities, splat, radii, point_list, ranges, depth, alpha = saved[:8]
        rest = saved[8:]
        opt = [rest.pop(0) if h else None for h in ctx.has]
        shs, colors_precomp, scales, rotations, cov3D_precomp = opt
        if not ctx.has_aux:
            raise RuntimeError("backward called but no input required grad in forward")
        aux = rest.pop(0)
        sh_jac = rest.pop(0) if ctx.has_jac else None
        group, views, counts = ctx.group, ctx.views, ctx.counts
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        dev = means3D.device
        n = means3D.shape[0]
        needs = list(ctx.needs_input_grad[:8])
        per_view_means2D = ctx.means2D_shape == (world, n, 3)
        with torch.cuda.device(dev):
            rows, ctx.rows = ctx.rows, None
            rows = R.render_backward(views[rank], splat, ranges, point_list, aux, depth, alpha, g_image, g_depth,
                                     g_alpha, rows, ctx.t_eps, ctx.band)
            total = int(sum(counts))
            # 4. every Gaussian's sums go back to its owner: (view, local Gaussian, one 64-byte gradient row of 16 floats)
            rows_mine = torch.empty((world * max(n, 1), R.grad_row_floats()), dtype=torch.float32, device=dev)[:world * n]
            with _stage("exchange_bwd"):
                w_rows = _all_to_all_rows(rows_mine, rows[:total], [n] * world, counts, group, True)
            rows_mine = rows_mine.view(world, n, R.grad_row_floats())
            # 5. chain rule through my shard's projection into each view, summed over the views.  My own view's rows
            # never travel: its launch reads them where render_backward left them and runs while the all-to-all is
            # in flight; the other world - 1 views then take ONE launch that adds their sum to it in place
            # (gs_preprocess_bwd_views with the accumulate flag: one read-modify-write of the gradients, not world - 1).
            off = int(sum(counts[:rank]))
            m2d_shape = (n, 3) if per_view_means2D else None
            acc = R.preprocess_backward(views[rank], means3D, opacities, scales, rotations, cov3D_precomp, shs,
                                        colors_precomp, radii[rank], rows[off:off + n], needs, m2d_shape,
                                        sh_jac=None if sh_jac is None else sh_jac[rank])
            g_m2d = {rank: acc[1]} if (per_view_means2D and needs[1]) else {}
            with _stage("exchange_bwd_wait"):
                _wait(w_rows)
            others = [u for u in range(world) if u != rank]
            if others:
                needs_o = list(needs)
                needs_o[1] = bool(per_view_means2D and needs[1])
                g = R.preprocess_backward_views([views[u] for u in others], means3D, opacities, scales, rotations,
                                                cov3D_precomp, shs, colors_precomp, [radii[u] for u in others],
                                                [rows_mine[u] for u in others], needs_o,
                                                sh_jacs=None if sh_jac is None else [sh_jac[u] for u in others],
                                                accumulate_into=acc)
                if needs_o[1]:
                    for j, u in enumerate(others):
                        g_m2d[u] = g[1][j]
            g_m2d = [g_m2d[v] for v in range(world)] if g_m2d else []
            acc = list(acc)
            if needs[1] and ctx.means2D_shape is not None:
                acc[1] = torch.stack(g_m2d) if per_view_means2D else \
                    torch.zeros(ctx.means2D_shape, dtype=torch.float32, device=dev)
        return (*acc, None, None, None, None, None)


class SplatExchange:
    """Training step for `world` ranks that each OWN a contiguous shard of the Gaussians (rank r holds
    global indices [sum(counts[:r]), sum(counts[:r+1]))) and each RENDER one view per step.

    No counterpart in the reference (single GPU, one view per iteration, train.py:129-135).  Instead of
    replicating the Gaussians and all-reducing 59 floats per Gaussian (GradReducer: 236 B x N through
    ring steps), the ranks exchange what the per-tile kernels actually consume and produce:

      forward   rank r projects its shard into every rank's view (gs_preprocess_fwd, `world` launches on
                N/world Gaussians each) and one all-to-all hands every renderer the 48-byte splat records,
                depth keys and boxes of all N Gaussians for its view (60 B per Gaussian and view);
      backward  the renderer's 64-byte gradient rows (the ten per-Gaussian sums of render_bwd) go back to
                the owners in one all-to-all, and each owner runs gs_preprocess_bwd per view on its shard.

    124 B per Gaussian and view, all of it direct peer-to-peer traffic -- the pattern the fully connected
    xGMI mesh is built for -- and parameters, gradients and optimizer state stay sharded (no replica,
    no parameter broadcast).  The image each rank gets is bit-identical to a single-GPU render of the
    concatenated Gaussians; the gradients equal the sum over the views of the single-GPU gradients.

    `means2D` (optional) has shape (world, n_local, 3): its gradient is the per-view screen-space
    gradient of the local Gaussians, so add_densification_stats (scene/gaussian_model.py:701-713) and
    max_radii2D need no collective at all (`radii` is returned per view: (world, n_local)).
    """

    def __init__(self, n_local: int, device=None, group: Optional[dist.ProcessGroup] = None):
        self.group = group
        self.set_local_count(n_local, device)

    def set_local_count(self, n_local: int, device=None):
        """(Re)publish the shard sizes; call again after densification / pruning changed n_local."""
        world = dist.get_world_size(self.group)
        use_dev = dist.get_backend(self.group) != "gloo"
        t = torch.tensor([int(n_local)], dtype=torch.int64, device=device if use_dev else None)
        allc = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(allc, t, group=self.group)
        self.counts = [int(c.item()) for c in allc]
        self.n_local = int(n_local)

    @property
    def offset(self) -> int:
        return sum(self.counts[:dist.get_rank(self.group)])

    def rasterize(self, views, means3D, means2D=None, shs=None, colors_precomp=None, opacities=None, scales=None,
                  rotations=None, cov3D_precomp=None, t_eps=None):
        """views: one GaussianRasterizationSettings per rank (views[r] is rendered by rank r; every rank
        passes the same list).  The other arguments are this rank's shard, as for GaussianRasterizer.
        Returns (image of views[rank], radii (world, n_local), depth, alpha)."""
        from . import rasterizer as R
        world = dist.get_world_size(self.group)
        if len(views) != world:
            raise ValueError(f"need one view per rank ({world}), got {len(views)}")
        if (shs is None) == (colors_precomp is None):
            raise ValueError("Please provide exactly one of either SHs or precomputed colors!")
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise ValueError("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
        if opacities is None:
            raise ValueError("opacities is required")
        n = means3D.shape[0]
        if n != self.n_local:
            raise ValueError(f"shard has {n} Gaussians, {self.n_local} were published (call set_local_count)")
        dev = means3D.device
        means3D = R._prep(means3D, "means3D", (3,), dev)
        shs = R._prep(shs, "shs", None, dev)
        colors_precomp = R._prep(colors_precomp, "colors_precomp", (3,), dev)
        opacities = R._prep(opacities.reshape(n, 1) if opacities.dim() == 1 else opacities, "opacities", (1,), dev)
        scales = R._prep(scales, "scales", (3,), dev)
        rotations = R._prep(rotations, "rotations", (4,), dev)
        cov3D_precomp = R._prep(cov3D_precomp, "cov3D_precomp", (6,), dev)
        need_grad = R.wants_grad(means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp)
        return _RasterizeSharded.apply(means3D, means2D, shs, colors_precomp, opacities, scales, rotations,
                                       cov3D_precomp, list(views), R.DEFAULT_T_EPS if t_eps is None else t_eps,
                                       list(self.counts), self.group, need_grad)


# ---------------------------------------------------------------------------------------------------
# View-sharded training step: Gaussians replicated, gradient all-reduce overlapped with the backward
# ---------------------------------------------------------------------------------------------------
def bucket_ranges(n: int, num_buckets: int, align: int = 256):
    """[begin, end) Gaussian ranges of at most num_buckets buckets; boundaries are multiples of `align` (whole
    workgroups of gs_preprocess_bwd) except the last end."""
    if n <= 0:
        return []
    per = -(-n // max(1, num_buckets))
    per = -(-per // align) * align
    return [(b, min(b + per, n)) for b in range(0, n, per)]


class _RasterizeViews(torch.autograd.Function):
    """k camera views of the same (replicated) Gaussians as ONE autograd node: the forward runs the per-Gaussian stage
    of all k views as one launch (gs_preprocess_fwd_views) and bins and composites view by view; the
    backward runs every view's tile backward, then the chain rule of ALL k views as one launch per Gaussian range
    (gs_preprocess_bwd_views: the views' sums never leave registers / LDS, no per-view gradient tensors, every gradient
    row written once) and all-reduces each range's gradients as soon as its launch is queued (see ViewShardedStep).
    Outputs: k images, k radii, k depths, k alphas."""

    @staticmethod
    def forward(ctx, means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp,
                views, t_eps, need_grad, owner):
        from . import rasterizer as R
        dev = means3D.device
        per_view = []
        N = means3D.shape[0]
        with torch.cuda.device(dev):
            # the per-Gaussian stage of ALL views in one launch (gs_preprocess_fwd_views: the SH coefficients, half of what
            # that stage reads, are read once), then binning and compositing view by view
            i32 = dict(dtype=torch.int32, device=dev)
            outs, jacs = [], []
            for _ in views:
                outs.append((torch.empty((max(N, 1), 12), dtype=torch.float32, device=dev)[:N],
                             torch.empty((2, max(N, 1)), **i32), torch.empty((2, max(N, 1)), **i32),
                             torch.empty((max(N, 1), 2), **i32)[:N]))
                jacs.append(torch.empty((N, R.sh_jac_row_floats()), dtype=torch.float32, device=dev)
                            if (need_grad and R.USE_SH_JAC and shs is not None and N > 0) else None)
            pre = R.preprocess_views(views, means3D, opacities, scales, rotations, cov3D_precomp, shs, colors_precomp,
                                     outs=[(o[0], o[1][0][:N], o[2][0][:N], o[3]) for o in outs],
                                     sh_jacs=jacs if jacs and jacs[0] is not None else None)
            for settings, o, jac, (splat, radii, tiles_touched, _, _, rects) in zip(views, outs, jacs, pre):
                band = R._band(settings, None)
                state = R.bin_splats(settings, splat, radii, tiles_touched, o[1], o[2], rects)._replace(sh_jac=jac)
                image, depth, alpha, aux, rows = R.render_forward(settings, state.splat, state.ranges, state.point_list,
                                                                  need_grad, t_eps, band)
                per_view.append((state, image, depth, alpha, aux, rows, band))
            del outs, pre
        k = len(views)
        ctx.k, ctx.views, ctx.t_eps, ctx.owner = k, list(views), float(t_eps), owner
        ctx.bands = [pv[6] for pv in per_view]
        ctx.rows = [pv[5] for pv in per_view]
        ctx.set_materialize_grads(False)
        ctx.has = tuple(t is not None for t in (shs, colors_precomp, scales, rotations, cov3D_precomp))
        ctx.means2D_shape = None if means2D is None else tuple(means2D.shape)
        saved = [means3D, opacities]
        saved += [t for t in (shs, colors_precomp, scales, rotations, cov3D_precomp) if t is not None]
        ctx.has_aux = need_grad
        ctx.has_jac = per_view[0][0].sh_jac is not None if k else False
        for state, image, depth, alpha, aux, rows, band in per_view:
            saved += [state.splat, state.radii, state.point_list, state.ranges, depth, alpha]
            if ctx.has_aux:
                saved.append(aux)
            if ctx.has_jac:
                saved.append(state.sh_jac)
        ctx.save_for_backward(*saved)
        radii = [pv[0].radii for pv in per_view]
        ctx.mark_non_differentiable(*radii)
        return (*[pv[1] for pv in per_view], *radii, *[pv[2] for pv in per_view], *[pv[3] for pv in per_view])

    @staticmethod
    def backward(ctx, *g_out):
        from . import rasterizer as R
        k, owner = ctx.k, ctx.owner
        g_images, g_depths, g_alphas = g_out[:k], g_out[2 * k:3 * k], g_out[3 * k:4 * k]
        saved = list(ctx.saved_tensors)
        means3D, opacities = saved[:2]
        rest = saved[2:]
        shs, colors_precomp, scales, rotations, cov3D_precomp = [rest.pop(0) if h else None for h in ctx.has]
        if not ctx.has_aux:
            raise RuntimeError("backward called but no input required grad in forward")
        per_view = []
        for _ in range(k):
            splat, radii, point_list, ranges, depth, alpha = rest[:6]
            del rest[:6]
            aux = rest.pop(0)
            jac = rest.pop(0) if ctx.has_jac else None
            per_view.append((splat, radii, point_list, ranges, depth, alpha, aux, jac))
        dev = means3D.device
        N = means3D.shape[0]
        needs = list(ctx.needs_input_grad[:8])
        f32 = dict(dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            # full-size destinations, in the order of GaussianRasterizer.forward's arguments; means2D's gradient is a
            # per-view statistic (densification): one (N, 3) slab per view, NOT summed over views or ranks
            per_row = (means3D, None, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp)
            full = [torch.empty_like(t) if (t is not None and needs[i]) else None for i, t in enumerate(per_row)]
            m2d_rows = False
            if needs[1] and ctx.means2D_shape is not None:
                m2d_rows = ctx.means2D_shape in ((k, N, 3), (N, 3)) and (k == 1 or ctx.means2D_shape == (k, N, 3))
                full[1] = torch.empty(ctx.means2D_shape, **f32) if m2d_rows else torch.zeros(ctx.means2D_shape, **f32)
            m2d_views = full[1].view(k, N, 3) if m2d_rows else None
            reduce_now = owner.world > 1 or owner.force_collectives
            works = []
            # lean launches: the camera tensors are converted once per view and the C-ABI is called on base pointers + row
            # offsets (per range the host must queue one kernel and one grouped all-reduce faster than the GPU finishes
            # them, or the overlap is lost to launch gaps)
            lib = R._lib.load()
            K = shs.shape[1] if shs is not None else 0
            st = R._stream()

            def addr(t, b):                    # address of row b (every per-Gaussian tensor here has 4-byte elements)
                return None if t is None else t.data_ptr() + b * t.stride(0) * 4

            def at(t, b):
                return None if t is None else R.C.c_void_p(addr(t, b))

            g_m3, _, g_sh, g_col, g_op, g_sc, g_rot, g_cov = full
            live = [v for v in range(k) if not (g_images[v] is None and g_depths[v] is None and g_alphas[v] is None)]
            if not live:                       # no view's outputs were used: zero gradients (still all-reduced below)
                for i, t in enumerate(full):
                    if t is not None:
                        t.zero_()
            elif m2d_views is not None and len(live) < k:
                m2d_views.zero_()
            # 1. every view's tile backward: ten sums per (Gaussian, view) in that view's own 64-byte rows
            descs = (R._lib.GsBwdView * max(len(live), 1))()
            keep = []                          # camera tensors stay alive until the launches are queued
            for j, v in enumerate(live):
                settings = ctx.views[v]
                splat, radii, point_list, ranges, depth, alpha, aux, jac = per_view[v]
                rows, ctx.rows[v] = ctx.rows[v], None          # cleared by the forward; a second backward clears its own
                rows = R.render_backward(settings, splat, ranges, point_list, aux, depth, alpha, g_images[v], g_depths[v],
                                         g_alphas[v], rows, ctx.t_eps, ctx.bands[v])
                cams = [R._camera_tensor(getattr(settings, name), n, name, dev) for name, n in
                        (("viewmatrix", 16), ("projmatrix", 16), ("campos", 3))]
                g_m2 = m2d_views[v] if m2d_views is not None else None
                jac = jac if shs is not None else None     # (N, 12): a range is a row slice like everything else
                keep.append((cams, rows, radii, jac, g_m2))
                d = descs[j]
                d.viewmatrix, d.projmatrix, d.campos = (t.data_ptr() for t in cams)
                d.H, d.W = settings.image_height, settings.image_width
            # 2. the chain rule per Gaussian range, ALL local views in one launch (gs_preprocess_bwd_views: the views' sums
            #    stay in registers / LDS and every gradient row is written once), each range followed by its all-reduce
            spans = bucket_ranges(N, owner.num_buckets) if reduce_now else ([(0, N)] if N else [])
            sh_degree = ctx.views[0].sh_degree
            for (b, e) in (spans if live else []):
                for j in range(len(live)):
                    _, rows, radii, jac, g_m2 = keep[j]
                    d = descs[j]
                    d.radii, d.grad_rows = addr(radii, b), addr(rows, b)
                    d.sh_jac, d.g_means2D = addr(jac, b), addr(g_m2, b)
                with _stage("preprocess_bwd"):
                    R._lib.check(lib.gs_preprocess_bwd_views(
                        len(live), descs, e - b, K, sh_degree, at(means3D, b), at(scales, b), at(rotations, b),
                        at(cov3D_precomp, b), at(opacities, b), at(shs, b), at(colors_precomp, b), at(g_m3, b),
                        at(g_sc, b), at(g_rot, b), at(g_cov, b), at(g_op, b), at(g_sh, b), at(g_col, b), 0, st),
                        "gs_preprocess_bwd_views")
                if reduce_now:   # queued behind this range's launch on RCCL's stream; the next range's kernel overlaps it
                    with _stage("allreduce_launch"):
                        works.append(_all_reduce_sum([t[b:e] for i, t in enumerate(full) if t is not None and i != 1],
                                                     owner.group, async_op=True))
            del keep
            if not live and reduce_now:       # keep the collective sequence identical on every rank
                for (b, e) in bucket_ranges(N, owner.num_buckets):
                    works.append(_all_reduce_sum([t[b:e] for i, t in enumerate(full) if t is not None and i != 1],
                                                 owner.group, async_op=True))
            with _stage("allreduce_wait"):
                for w in works:
                    _wait(w)
            if owner.average and owner.world > 1:
                for i, t in enumerate(full):
                    if t is not None and i != 1:
                        t.div_(owner.world)
        return (*full, None, None, None, None)


class ViewShardedStep:
    """One training step of the pattern BASELINE.json's north_star names (6 M Gaussians, 8 camera
    views per step): every rank holds ALL Gaussians and renders ITS views of the step -- shard_views(num_views, rank,
    world), i.e. 8 / world of them -- and the one exchange of the step is the SUM of the per-Gaussian gradients (59 floats
    per Gaussian at SH degree 3).  No counterpart in the reference (one view per iteration on one device,
    train.py:129-135); equal to sequential accumulation of the same views on one device up to fp32 summation order.

    The backward first runs the tile backward of every local view (each into its own 64-byte gradient rows), then the
    per-Gaussian chain rule of ALL local views at once, one gs_preprocess_bwd_views launch per Gaussian range -- the views'
    sums are formed in registers / LDS in view order and every gradient row is written once (views 1..k-1 of the rank cost
    no gradient tensors, no read-modify-write pass and no collective); the C-ABI takes base pointers and a count, so a
    range is a row slice of every per-Gaussian tensor -- and launches that range's all-reduce (one grouped RCCL call over
    its slices of the gradient tensors) right after its kernel is queued: RCCL's stream starts on range b while this
    stream computes range b + 1, and only the last range's transfer is exposed.  One exchange per step, whatever the number
    of local views: the wire time is amortised over 8 / world views.  The forward runs the per-Gaussian stage of all local
    views as one launch too (gs_preprocess_fwd_views: the SH coefficients are read once per step, not once per view).
    The gradients autograd hands to the parameters are already summed over views and ranks
    (divided by the world size if `average`).  means2D (optional, (k, N, 3)): its gradient is the per-view screen-space
    gradient (the densification statistic); it and radii (k of them) are not exchanged -- fold them into the per-Gaussian
    statistics view by view, then call reduce_densify_stats ONCE per step.

        step = ViewShardedStep([settings of my views], num_buckets=4)
        images, radii, depths, alphas = step(means3D=..., means2D=zeros(k, N, 3), shs=..., ...)   # lists of k
        sum(loss(images[j], gt[j]) for j in range(k)).backward()
    """

    def __init__(self, views, t_eps=None, num_buckets: int = 4, average: bool = False,
                 group: Optional[dist.ProcessGroup] = None, force_collectives: bool = False):
        from .rasterizer import DEFAULT_T_EPS
        self.views = list(views)
        if not self.views:
            raise ValueError("ViewShardedStep needs at least one view per rank (give every rank a view, or a smaller world)")
        if len({int(rs.sh_degree) for rs in self.views}) > 1:
            raise ValueError("the views of one step must share sh_degree (one model, one active degree)")
        self.t_eps = DEFAULT_T_EPS if t_eps is None else t_eps
        self.num_buckets = int(num_buckets)
        self.average = average
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # world 1 has nothing to exchange; force_collectives issues the (self-)all-reduces anyway (bench.py --force-dist)
        self.force_collectives = bool(force_collectives) and dist.is_initialized()

    def __call__(self, means3D, means2D=None, shs=None, colors_precomp=None, opacities=None, scales=None,
                 rotations=None, cov3D_precomp=None, cov3Ds_precomp=None):
        from . import rasterizer as R
        k = len(self.views)
        m3 = sh = col = op = sc = rot = cov = None
        for rs in self.views:     # the argument checks of GaussianRasterizer.forward, against every view's settings
            m3, _, sh, col, op, sc, rot, cov = R.check_inputs(rs, means3D, None, shs, colors_precomp, opacities, scales,
                                                               rotations, cov3D_precomp, cov3Ds_precomp)
        N = m3.shape[0]
        if means2D is not None and tuple(means2D.shape) != (k, N, 3) and not (k == 1 and tuple(means2D.shape) == (N, 3)):
            raise ValueError(f"means2D must be ({k}, {N}, 3): one screen-space gradient per local view")
        need_grad = R.wants_grad(m3, means2D, sh, col, op, sc, rot, cov)
        out = _RasterizeViews.apply(m3, means2D, sh, col, op, sc, rot, cov, self.views, self.t_eps, need_grad, self)
        return list(out[:k]), list(out[k:2 * k]), list(out[2 * k:3 * k]), list(out[3 * k:4 * k])


class ViewShardedRasterizer(ViewShardedStep):
    """ViewShardedStep with ONE view per rank and step, as a drop-in for GaussianRasterizer(raster_settings): same call
    signature, same 4-tuple; the backward's per-range gs_preprocess_bwd launches overlap the bucketed all-reduce."""

    def __init__(self, raster_settings, t_eps=None, num_buckets: int = 4, average: bool = False,
                 group: Optional[dist.ProcessGroup] = None, force_collectives: bool = False):
        super().__init__([raster_settings], t_eps, num_buckets, average, group, force_collectives)
        self.raster_settings = raster_settings

    def __call__(self, *args, **kwargs):
        images, radii, depths, alphas = super().__call__(*args, **kwargs)
        return images[0], radii[0], depths[0], alphas[0]
