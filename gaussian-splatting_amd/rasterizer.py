"""Host side of the MI355X Gaussian rasterizer: the reference's operator interface on top of the C-ABI.

Mirrors the two classes the reference imports from ``diff_gaussian_rasterization``
(reference gaussian_renderer/__init__.py:19-27) and calls at :563-638:

    GaussianRasterizationSettings(image_height, image_width, tanfovx, tanfovy, bg, scale_modifier,
                                  viewmatrix, projmatrix, sh_degree, campos, prefiltered, debug
                                  [, antialiasing])
    GaussianRasterizer(raster_settings)(means3D, means2D, shs, colors_precomp, opacities, scales,
                                        rotations, cov3D_precomp)  ->  (image, radii, depth, alpha)

Semantics are those of the reference's built-in rasterizer (:64-442), which is the parity oracle:
``tanfovx/tanfovy/scale_modifier/prefiltered/debug/antialiasing`` are stored but unused there
 and therefore here too.

PyTorch is used for device memory, the current stream and autograd plumbing only; every stage
runs in libgsplat_hip.so (hand-written HIP for gfx950).  No CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os
import threading
from typing import NamedTuple, Optional

import torch
from torch import nn

from . import _lib
from .profiling import stage as _stage

TILE = 16
_grad_row = None


def grad_row_floats() -> int:
    """Floats per row of the rasterizer's gradient buffer: the library's build constant (gs_grad_row_floats)."""
    global _grad_row
    if _grad_row is None:
        _grad_row = int(_lib.load().gs_grad_row_floats())
    return _grad_row

# Per-pixel stop threshold on the transmittance 1 - alpha_acc (see csrc/render.hip).  0.0 disables
# it (the reference's "no early termination"); what a pixel drops at the default is at most
# 1e-5 * max|colour| (1e-5 * max depth for the depth map), a tenth of the stated 1e-4 image / 1e-3
# depth tolerance; gradients lose terms of the same relative size.
# GSPLAT_T_EPS in the environment (read once, at import) replaces the default for callers that cannot pass t_eps --
# the reference's adapter constructs GaussianRasterizer(raster_settings=...) and nothing else: GSPLAT_T_EPS=0 gives
# it the reference's exact compositing.
def _default_t_eps() -> float:
    v = os.environ.get("GSPLAT_T_EPS")
    if v is None or v.strip() == "":
        return 1e-5
    t = float(v)
    if not (0.0 <= t < 1.0):
        raise ValueError(f"GSPLAT_T_EPS={v!r}: expected a transmittance threshold in [0, 1)")
    return t


DEFAULT_T_EPS = _default_t_eps()


class GaussianRasterizationSettings:
    """Value holder with the fields of reference gaussian_renderer/__init__.py:30-61 (+ antialiasing, :576)."""

    def __init__(self, image_height, image_width, tanfovx, tanfovy, bg, scale_modifier, viewmatrix,
                 projmatrix, sh_degree, campos, prefiltered, debug, antialiasing=False):
        self.image_height = int(image_height)
        self.image_width = int(image_width)
        self.tanfovx = tanfovx
        self.tanfovy = tanfovy
        self.bg = bg
        self.scale_modifier = scale_modifier
        self.viewmatrix = viewmatrix
        self.projmatrix = projmatrix
        self.sh_degree = int(sh_degree)
        self.campos = campos
        self.prefiltered = prefiltered
        self.debug = debug
        self.antialiasing = antialiasing


class BinningState(NamedTuple):
    """Everything the forward leaves in HBM for the backward / for inspection."""
    splat: torch.Tensor          # (N,12) f32 splat records (see csrc/gsplat_common.h)
    radii: torch.Tensor          # (N,) i32
    tiles_touched: torch.Tensor  # (N,) u32 as i32 storage
    order: torch.Tensor          # (N,) Gaussian ids far-to-near (culled ones last), stable in the index
    offsets: torch.Tensor        # (N,) exclusive scan of tiles_touched[order]
    num_pairs: int               # D
    keys: torch.Tensor           # (D,) tile id per sorted pair (int32 storage)
    point_list: torch.Tensor     # (D,) Gaussian index per sorted pair (int32 storage)
    ranges: torch.Tensor         # (tiles,2) i32
    sh_jac: Optional[torch.Tensor] = None   # (N,12) f32: the forward's colour side table for the backward (gsplat.h)


def _ptr(t: Optional[torch.Tensor]):
    if t is None:
        return None
    p = t.data_ptr()
    if p == 0 and t.untyped_storage().nbytes() > 0:   # zero-row view of a real allocation (N == 0)
        p = t.untyped_storage().data_ptr()
    return C.c_void_p(p)


def _prep(t: Optional[torch.Tensor], name: str, shape_tail, device) -> Optional[torch.Tensor]:
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError(f"{name} must live on the GPU (got {t.device}); this back-end has no CPU path")
    if t.device != device:
        raise RuntimeError(f"{name} is on {t.device}, expected {device}")
    if t.dtype != torch.float32:
        raise RuntimeError(f"{name} must be float32 (got {t.dtype})")
    if shape_tail is not None and tuple(t.shape[1:]) != tuple(shape_tail):
        raise ValueError(f"{name} has shape {tuple(t.shape)}, expected (N,{','.join(map(str, shape_tail))})")
    return t.contiguous()


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _camera_tensor(t, n, name, device):
    if not torch.is_tensor(t):
        t = torch.as_tensor(t, dtype=torch.float32)
    t = t.detach().to(device=device, dtype=torch.float32).contiguous()
    if t.numel() != n:
        raise ValueError(f"{name} must have {n} elements (got {tuple(t.shape)})")
    return t


def _band(settings, tile_rows):
    tiles_y = (settings.image_height + TILE - 1) // TILE
    if tile_rows is None:
        return 0, tiles_y
    b, e = int(tile_rows[0]), int(tile_rows[1])
    if not (0 <= b <= e <= tiles_y):
        raise ValueError(f"tile_rows {tile_rows} outside 0..{tiles_y}")
    return b, e


def sh_jac_row_floats() -> int:
    """Floats per row of the forward's colour side table (gs_sh_jac_row_floats)."""
    return int(_lib.load().gs_sh_jac_row_floats())


def _jac_ptr(sh_jac, N):
    """Pointer of the colour side table for the C-ABI; NULL without one."""
    if sh_jac is None:
        return None
    if sh_jac.dtype != torch.float32 or tuple(sh_jac.shape) != (N, sh_jac_row_floats()) or not sh_jac.is_contiguous():
        raise ValueError(f"sh_jac must be a contiguous float32 (N,{sh_jac_row_floats()}) tensor")
    return _ptr(sh_jac)


def preprocess(settings: GaussianRasterizationSettings, means3D, opacities, scales, rotations, cov3D_precomp, shs,
               colors_precomp, tile_rows=None, out=None, sh_jac=None):
    """gs_preprocess_fwd on the current stream.  Returns (splat (N,12), radii, tiles_touched, dkeys (N,),
    ids (N,), rects (N,2)).  `out` = preallocated contiguous (splat, dkeys, ids, rects) to write into
    (rows of the sort's ping-pong buffers, or slices of an exchange buffer: parallel.SplatExchange).
    sh_jac = a (N, gs_sh_jac_row_floats()) float32 tensor to receive the colour side table (shs only)."""
    lib = _lib.load()
    dev = means3D.device
    N = means3D.shape[0]
    H, W = settings.image_height, settings.image_width
    K = shs.shape[1] if shs is not None else 0
    view = _camera_tensor(settings.viewmatrix, 16, "viewmatrix", dev)
    proj = _camera_tensor(settings.projmatrix, 16, "projmatrix", dev)
    campos = _camera_tensor(settings.campos, 3, "campos", dev)
    row_b, row_e = _band(settings, tile_rows)
    i32 = dict(dtype=torch.int32, device=dev)
    if out is None:
        out = (torch.empty((max(N, 1), 12), dtype=torch.float32, device=dev)[:N], torch.empty((max(N, 1),), **i32)[:N],
               torch.empty((max(N, 1),), **i32)[:N], torch.empty((max(N, 1), 2), **i32)[:N])
    splat, dkeys, ids, rects = out
    if (tuple(splat.shape), tuple(dkeys.shape), tuple(ids.shape), tuple(rects.shape)) != ((N, 12), (N,), (N,), (N, 2)) \
            or not all(t.is_contiguous() for t in out):
        raise ValueError("preprocess: output buffers must be contiguous (N,12), (N,), (N,), (N,2)")
    radii = torch.empty((N,), **i32)
    tiles_touched = torch.empty((N,), **i32)
    jac_p = _jac_ptr(sh_jac if shs is not None else None, N)
    with _stage("preprocess_fwd"):
        _lib.check(lib.gs_preprocess_fwd(N, K, settings.sh_degree, H, W, _ptr(view), _ptr(proj), _ptr(campos),
                                         _ptr(means3D), _ptr(scales), _ptr(rotations), _ptr(cov3D_precomp),
                                         _ptr(opacities), _ptr(shs), _ptr(colors_precomp), _ptr(splat),
                                         _ptr(radii), _ptr(tiles_touched), _ptr(dkeys), _ptr(ids),
                                         _ptr(rects), row_b, row_e, jac_p, _stream()), "gs_preprocess_fwd")
    return splat, radii, tiles_touched, dkeys, ids, rects


def preprocess_views(views, means3D, opacities, scales, rotations, cov3D_precomp, shs, colors_precomp, outs=None,
                     sh_jacs=None):
    """gs_preprocess_fwd_views on the current stream: `preprocess` for k views of the same Gaussians in one launch (the
    per-Gaussian inputs, above all the SH coefficients, are read once).  views = k settings; outs = k tuples
    (splat, dkeys, ids, rects) as in `preprocess` (allocated here if None); sh_jacs = k side tables or None.
    Returns k tuples (splat, radii, tiles_touched, dkeys, ids, rects)."""
    lib = _lib.load()
    dev = means3D.device
    N, k = means3D.shape[0], len(views)
    K = shs.shape[1] if shs is not None else 0
    if len({int(st.sh_degree) for st in views}) > 1:
        raise ValueError("the views of one call must share sh_degree")
    i32 = dict(dtype=torch.int32, device=dev)
    descs = (_lib.GsFwdView * max(k, 1))()
    keep, res = [], []
    for j, st in enumerate(views):
        out = outs[j] if outs is not None else (
            torch.empty((max(N, 1), 12), dtype=torch.float32, device=dev)[:N], torch.empty((max(N, 1),), **i32)[:N],
            torch.empty((max(N, 1),), **i32)[:N], torch.empty((max(N, 1), 2), **i32)[:N])
        splat, dkeys, ids, rects = out
        if (tuple(splat.shape), tuple(dkeys.shape), tuple(ids.shape), tuple(rects.shape)) != ((N, 12), (N,), (N,), (N, 2)) \
                or not all(t.is_contiguous() for t in out):
            raise ValueError("preprocess_views: output buffers must be contiguous (N,12), (N,), (N,), (N,2)")
        radii = torch.empty((N,), **i32)
        tiles_touched = torch.empty((N,), **i32)
        cams = [_camera_tensor(getattr(st, name), n, name, dev) for name, n in
                (("viewmatrix", 16), ("projmatrix", 16), ("campos", 3))]
        keep.append(cams)
        jac = sh_jacs[j] if (sh_jacs is not None and shs is not None) else None
        d = descs[j]
        d.viewmatrix, d.projmatrix, d.campos = (t.data_ptr() for t in cams)
        d.splat, d.radii, d.tiles_touched = splat.data_ptr(), radii.data_ptr(), tiles_touched.data_ptr()
        d.depth_keys, d.ids, d.rects = dkeys.data_ptr(), ids.data_ptr(), rects.data_ptr()
        d.sh_jac = _jac_ptr(jac, N).value if jac is not None else None
        d.H, d.W = st.image_height, st.image_width
        d.tile_row_begin, d.tile_row_end = _band(st, None)
        res.append((splat, radii, tiles_touched, dkeys, ids, rects))
    with _stage("preprocess_fwd"):
        _lib.check(lib.gs_preprocess_fwd_views(k, descs, N, K, views[0].sh_degree if k else 0, _ptr(means3D), _ptr(scales),
                                               _ptr(rotations), _ptr(cov3D_precomp), _ptr(opacities), _ptr(shs),
                                               _ptr(colors_precomp), _stream()), "gs_preprocess_fwd_views")
    return res


# the forward leaves d colour / d dir for the backward (one backward launch that never re-reads the coefficients);
# False = the two-launch backward that does (kept for callers of the C-ABI that pass no table)
USE_SH_JAC = True


def preprocess_and_bin(settings: GaussianRasterizationSettings, means3D, opacities, scales, rotations,
                       cov3D_precomp, shs, colors_precomp, tile_rows=None, want_jac=False) -> BinningState:
    """preprocess -> depth sort -> scan -> duplicate_with_keys -> tile sort -> tile ranges (current stream).
    want_jac: a backward will follow -- with shs, also write the colour side table (BinningState.sh_jac)."""
    dev = means3D.device
    N = means3D.shape[0]
    i32 = dict(dtype=torch.int32, device=dev)
    # depth keys / Gaussian ids: ping-pong buffers of the per-Gaussian depth sort, input in row 0
    dkeys = torch.empty((2, max(N, 1)), **i32)
    ids = torch.empty((2, max(N, 1)), **i32)
    splat = torch.empty((max(N, 1), 12), dtype=torch.float32, device=dev)[:N]
    rects = torch.empty((max(N, 1), 2), **i32)[:N]
    sh_jac = None
    if want_jac and USE_SH_JAC and shs is not None and N > 0:
        sh_jac = torch.empty((N, sh_jac_row_floats()), dtype=torch.float32, device=dev)
    _, radii, tiles_touched, _, _, _ = preprocess(settings, means3D, opacities, scales, rotations, cov3D_precomp, shs,
                                                  colors_precomp, tile_rows, out=(splat, dkeys[0][:N], ids[0][:N], rects),
                                                  sh_jac=sh_jac)
    return bin_splats(settings, splat, radii, tiles_touched, dkeys, ids, rects, tile_rows)._replace(sh_jac=sh_jac)


def bin_splats(settings, splat, radii, tiles_touched, dkeys, ids, rects, tile_rows=None, before_scan=None) -> BinningState:
    """depth sort -> scan -> duplicate_with_keys -> tile sort -> tile ranges for N preprocessed Gaussians.
    dkeys / ids: (2,N) ping-pong buffers with the sort input in row 0; rects: (N,2).  `splat` itself is not read
    here (only carried into the result).  before_scan: called after the depth sort has been queued and before
    the first kernel that reads `rects` (parallel.SplatExchange waits for the boxes' transfer there)."""
    lib = _lib.load()
    dev = splat.device
    N = splat.shape[0]
    H, W = settings.image_height, settings.image_width
    st = _stream()
    row_b, row_e = _band(settings, tile_rows)
    i32 = dict(dtype=torch.int32, device=dev)
    in_b = C.c_int(0)
    ws_bytes = lib.gs_sort_workspace_bytes(N)
    sort_ws = torch.empty((max(16, ws_bytes),), dtype=torch.uint8, device=dev)
    with _stage("sort_depth"):
        _lib.check(lib.gs_sort_pairs(N, None, 0, 32, _ptr(dkeys[0]), _ptr(ids[0]), _ptr(dkeys[1]), _ptr(ids[1]),
                                       _ptr(sort_ws), ws_bytes, C.byref(in_b), st), "gs_sort_pairs")
    order = ids[1 if in_b.value else 0][:N]
    sorted_dkeys = dkeys[1 if in_b.value else 0][:N] if SKIP_CULLED_RANKS else None     # 0xFFFFFFFF = not drawn: its box is not fetched
    if before_scan is not None:
        before_scan()
    offsets = torch.empty((N,), **i32)
    rect_in_order = torch.empty((max(N, 1), 2), **i32)
    total = torch.empty((1,), **i32)
    scan_ws = torch.empty((max(16, lib.gs_scan_workspace_bytes(N)),), dtype=torch.uint8, device=dev)
    tiles_x, tiles_y = (W + TILE - 1) // TILE, (H + TILE - 1) // TILE
    ntiles = tiles_x * tiles_y

    def bin_pairs(cap, count_dev):
        """duplicate -> tile sort -> ranges into buffers of `cap` pairs; the pair count is `cap` itself
        (count_dev None) or read by the kernels from device memory (no host round trip)."""
        keys_a = torch.empty((max(cap, 1),), **i32)
        keys_b = torch.empty((max(cap, 1),), **i32)
        vals_a = torch.empty((max(cap, 1),), **i32)
        vals_b = torch.empty((max(cap, 1),), **i32)
        with _stage("duplicate_with_keys"):
            _lib.check(lib.gs_duplicate_with_keys(N, W, _ptr(rect_in_order), _ptr(order), _ptr(offsets),
                                                  _ptr(keys_a), _ptr(vals_a), cap, row_b, row_e, st),
                       "gs_duplicate_with_keys")
        # pairs were emitted far-to-near, so only the tile id is left to sort (stable)
        bits = max(1, (ntiles - 1).bit_length())
        ws_bytes = lib.gs_sort_workspace_bytes(cap)
        sort_ws = torch.empty((max(16, ws_bytes),), dtype=torch.uint8, device=dev)
        with _stage("sort_pairs"):
            _lib.check(lib.gs_sort_pairs(cap, _ptr(count_dev), 0, bits, _ptr(keys_a), _ptr(vals_a), _ptr(keys_b),
                                         _ptr(vals_b), _ptr(sort_ws), ws_bytes, C.byref(in_b), st), "gs_sort_pairs")
        keys, point_list = (keys_b, vals_b) if in_b.value else (keys_a, vals_a)
        ranges = torch.empty((ntiles, 2), **i32)
        with _stage("identify_tile_ranges"):
            _lib.check(lib.gs_identify_tile_ranges(cap, _ptr(count_dev), ntiles, _ptr(keys), _ptr(ranges), st),
                       "gs_identify_tile_ranges")
        return keys, point_list, ranges

    # The pair count D is only known on the device after the scan.  Reading it back before launching
    # the rest stalls the GPU for ~0.1 ms per frame, so once a capacity is known from earlier frames
    # on this device the rest is launched against that capacity with D taken from device memory,
    # and D is checked afterwards (it arrived long before: the copy was queued right after the scan).
    cap_key = (dev.index, tiles_x, tiles_y, row_b, row_e)
    cap = _pair_capacity.get(cap_key) if OPTIMISTIC_BINNING else 0
    D = None
    out = None
    if cap > 0:
        with _stage("scan_tiles"):
            _lib.check(lib.gs_scan_tiles(N, _ptr(rects), _ptr(order), _ptr(sorted_dkeys), _ptr(rect_in_order),
                                         _ptr(offsets), _ptr(total), _ptr(scan_ws), None, row_b, row_e, st),
                       "gs_scan_tiles")
        pinned, event = _readback_slot(dev)
        pinned.copy_(total, non_blocking=True)
        event.record()
        out = bin_pairs(cap, total)
        event.synchronize()
        D = int(pinned.item()) & 0xFFFFFFFF
        if D > cap:        # estimate too small: pairs were dropped, redo with the exact count
            out = None
    else:
        host_total = C.c_uint32(0)
        with _stage("scan_tiles"):
            _lib.check(lib.gs_scan_tiles(N, _ptr(rects), _ptr(order), _ptr(sorted_dkeys), _ptr(rect_in_order),
                                         _ptr(offsets), _ptr(total), _ptr(scan_ws), C.byref(host_total), row_b,
                                         row_e, st), "gs_scan_tiles")
        D = int(host_total.value)
    if out is None:
        out = bin_pairs(D, None)
    _pair_capacity.learn(cap_key, D)
    keys, point_list, ranges = out
    return BinningState(splat, radii, tiles_touched, order, offsets, D, keys[:D], point_list[:D], ranges)


# gs_scan_tiles gets the sorted depth keys and skips the box gather of ranks that are not drawn (False: gathers every rank)
SKIP_CULLED_RANKS = True

# per-device state of the optimistic binning: pair capacity learnt from earlier frames, and a pinned
# 4-byte slot + event for the asynchronous read-back of D
OPTIMISTIC_BINNING = True


class PairCapacityCache:
    """Pair capacity learnt from earlier frames, per (device, tile grid, band of tile rows): two image sizes -- or two
    threads rendering different cameras -- on one device do not overwrite each other's estimate (a capacity that is
    too small costs a second binning pass, one that is too large only memory).  Guarded by a lock (SURVEY.md 8b:
    the only mutable state of the host side is a per-device cache behind a mutex)."""

    def __init__(self):
        self._lock = threading.Lock()
        self._cap = {}

    def get(self, key) -> int:
        with self._lock:
            return self._cap.get(key, 0)

    def learn(self, key, num_pairs: int) -> int:
        with self._lock:
            cap = max(self._cap.get(key, 0), num_pairs + num_pairs // 4 + 4096)
            self._cap[key] = cap
            return cap

    def clear(self):
        with self._lock:
            self._cap.clear()

    def snapshot(self) -> dict:
        with self._lock:
            return dict(self._cap)

    def restore(self, snap: dict):
        with self._lock:
            self._cap = dict(snap)


_pair_capacity = PairCapacityCache()
_readback = {}
_readback_lock = threading.Lock()


def _readback_slot(dev):
    """One pinned slot + event per (device, stream, thread): frames queued on different streams or from different
    threads do not share it."""
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream, threading.get_ident())
    with _readback_lock:
        slot = _readback.get(key)
        if slot is None:
            slot = (torch.empty((1,), dtype=torch.int32, pin_memory=True), torch.cuda.Event())
            _readback[key] = slot
    return slot


def render_forward(settings, splat, ranges, point_list, need_grad, t_eps, band):
    """gs_render_fwd on the current stream -> (image, depth, alpha, aux, rows); aux / rows (the backward's
    per-Gaussian gradient rows, cleared by the VALU-bound forward kernel for free) only if need_grad."""
    lib = _lib.load()
    dev = splat.device
    N = splat.shape[0]
    H, W = settings.image_height, settings.image_width
    bg = _camera_tensor(settings.bg, 3, "bg", dev)
    f32 = dict(dtype=torch.float32, device=dev)
    image = torch.empty((3, H, W), **f32)
    depth = torch.empty((1, H, W), **f32)
    alpha = torch.empty((1, H, W), **f32)
    aux = torch.empty((3, H, W), **f32) if need_grad else None
    rows = torch.empty((max(N, 1), grad_row_floats()), **f32) if need_grad else None
    with _stage("render_fwd"):
        _lib.check(lib.gs_render_fwd(H, W, _ptr(bg), _ptr(splat), _ptr(ranges), _ptr(point_list), _ptr(image),
                                     _ptr(depth), _ptr(alpha), _ptr(aux), float(t_eps), band[0], band[1], _ptr(rows),
                                     rows.numel() * 4 if rows is not None else 0, _stream()), "gs_render_fwd")
    return image, depth, alpha, aux, rows


def render_backward(settings, splat, ranges, point_list, aux, depth, alpha, g_image, g_depth, g_alpha, rows, t_eps, band):
    """gs_render_bwd on the current stream: adds this frame's per-Gaussian sums into `rows` ((N,16), zeroed
    by the forward; None = allocate zeros) and returns it."""
    lib = _lib.load()
    dev = splat.device
    N = splat.shape[0]
    H, W = settings.image_height, settings.image_width

    def gin(t, shape):
        if t is None:
            return None
        return t.to(torch.float32).expand(shape).contiguous()

    g_image = gin(g_image, (3, H, W))
    if g_image is None:
        g_image = torch.zeros((3, H, W), dtype=torch.float32, device=dev)
    g_depth = gin(g_depth, (1, H, W))
    g_alpha = gin(g_alpha, (1, H, W))
    if rows is None:
        rows = torch.zeros((max(N, 1), grad_row_floats()), dtype=torch.float32, device=dev)
    with _stage("render_bwd"):
        _lib.check(lib.gs_render_bwd(H, W, _ptr(splat), _ptr(ranges), _ptr(point_list), _ptr(aux), _ptr(depth),
                                     _ptr(alpha), _ptr(g_image), _ptr(g_depth), _ptr(g_alpha), _ptr(rows), float(t_eps),
                                     band[0], band[1], _stream()), "gs_render_bwd")
    return rows


def preprocess_backward(settings, means3D, opacities, scales, rotations, cov3D_precomp, shs, colors_precomp, radii,
                        rows, needs, means2D_shape, accumulate_into=None, out=None, sh_jac=None):
    """gs_preprocess_bwd on the current stream.  `needs` = 8 flags in the order of GaussianRasterizer.forward's
    arguments; returns the 8 gradients (None where not needed).  accumulate_into = the 8-tuple a previous
    call returned: this view's gradients are added into those tensors (means2D's, per view, is fresh).
    out = an 8-tuple of preallocated destinations (None where not needed) that this call OVERWRITES: row
    slices of full-size gradient tensors when the chain rule runs per Gaussian range (every per-Gaussian
    argument is then the same row slice; parallel.ViewShardedRasterizer).
    sh_jac = the side table the forward wrote for these Gaussians and this view (BinningState.sh_jac), or None."""
    lib = _lib.load()
    dev = means3D.device
    N = means3D.shape[0]
    H, W = settings.image_height, settings.image_width
    K = shs.shape[1] if shs is not None else 0
    view = _camera_tensor(settings.viewmatrix, 16, "viewmatrix", dev)
    proj = _camera_tensor(settings.projmatrix, 16, "projmatrix", dev)
    campos = _camera_tensor(settings.campos, 3, "campos", dev)
    f32 = dict(dtype=torch.float32, device=dev)
    g_means2D = torch.empty(means2D_shape, **f32) if (needs[1] and means2D_shape == (N, 3)) else None
    if accumulate_into is not None:
        g_means3D, _, g_shs, g_colors, g_opac, g_scales, g_rots, g_cov = accumulate_into
    elif out is not None:
        g_means3D, o_m2d, g_shs, g_colors, g_opac, g_scales, g_rots, g_cov = out
        if o_m2d is not None:
            g_means2D = o_m2d
        for t in out:
            if t is not None and (not t.is_contiguous() or t.shape[0] != N or t.dtype != torch.float32):
                raise ValueError("preprocess_backward: out tensors must be contiguous float32 with N rows")
    else:
        g_means3D = torch.empty((N, 3), **f32) if needs[0] else None
        g_shs = torch.empty_like(shs) if (shs is not None and needs[2]) else None
        g_colors = torch.empty((N, 3), **f32) if (colors_precomp is not None and needs[3]) else None
        g_opac = torch.empty((N, 1), **f32) if needs[4] else None
        g_scales = torch.empty((N, 3), **f32) if (scales is not None and needs[5]) else None
        g_rots = torch.empty((N, 4), **f32) if (rotations is not None and needs[6]) else None
        g_cov = torch.empty((N, 6), **f32) if (cov3D_precomp is not None and needs[7]) else None
    jac_p = _jac_ptr(sh_jac if shs is not None else None, N)
    with _stage("preprocess_bwd"):
        _lib.check(lib.gs_preprocess_bwd(N, K, settings.sh_degree, H, W, _ptr(view), _ptr(proj), _ptr(campos),
                                         _ptr(means3D), _ptr(scales), _ptr(rotations), _ptr(cov3D_precomp),
                                         _ptr(opacities), _ptr(shs), _ptr(colors_precomp), _ptr(radii), _ptr(rows),
                                         _ptr(g_means3D), _ptr(g_means2D), _ptr(g_scales), _ptr(g_rots),
                                         _ptr(g_cov), _ptr(g_opac), _ptr(g_shs), _ptr(g_colors),
                                         0 if accumulate_into is None else 1, jac_p, _stream()),
                   "gs_preprocess_bwd")
    if needs[1] and g_means2D is None and means2D_shape is not None:
        g_means2D = torch.zeros(means2D_shape, **f32)
    return g_means3D, g_means2D, g_shs, g_colors, g_opac, g_scales, g_rots, g_cov


def preprocess_backward_views(views, means3D, opacities, scales, rotations, cov3D_precomp, shs, colors_precomp, radii,
                              rows, needs, sh_jacs=None, accumulate_into=None):
    """gs_preprocess_bwd_views on the current stream: the chain rule of k views of the same Gaussians, summed over the
    views in view order, every gradient row written once.  views = k settings; radii, rows, sh_jacs = k tensors each (what
    each view's forward / tile backward left).  Returns the 8 gradients in the order of GaussianRasterizer.forward's
    arguments; means2D's is (k, N, 3), one slab per view (a per-view statistic, never summed).  accumulate_into = the
    8-tuple an earlier call returned: the sums are added into those tensors."""
    lib = _lib.load()
    dev = means3D.device
    N, k = means3D.shape[0], len(views)
    K = shs.shape[1] if shs is not None else 0
    if len({int(st.sh_degree) for st in views}) > 1:
        raise ValueError("the views of one call must share sh_degree")
    f32 = dict(dtype=torch.float32, device=dev)
    g_means2D = torch.empty((k, N, 3), **f32) if needs[1] else None
    if accumulate_into is not None:
        g_means3D, _, g_shs, g_colors, g_opac, g_scales, g_rots, g_cov = accumulate_into
    else:
        g_means3D = torch.empty((N, 3), **f32) if needs[0] else None
        g_shs = torch.empty_like(shs) if (shs is not None and needs[2]) else None
        g_colors = torch.empty((N, 3), **f32) if (colors_precomp is not None and needs[3]) else None
        g_opac = torch.empty((N, 1), **f32) if needs[4] else None
        g_scales = torch.empty((N, 3), **f32) if (scales is not None and needs[5]) else None
        g_rots = torch.empty((N, 4), **f32) if (rotations is not None and needs[6]) else None
        g_cov = torch.empty((N, 6), **f32) if (cov3D_precomp is not None and needs[7]) else None
    descs = (_lib.GsBwdView * max(k, 1))()
    keep = []
    for j, st in enumerate(views):
        cams = [_camera_tensor(getattr(st, name), n, name, dev) for name, n in
                (("viewmatrix", 16), ("projmatrix", 16), ("campos", 3))]
        keep.append(cams)
        d = descs[j]
        d.viewmatrix, d.projmatrix, d.campos = (t.data_ptr() for t in cams)
        d.radii, d.grad_rows = radii[j].data_ptr(), rows[j].data_ptr()
        jac = sh_jacs[j] if (sh_jacs is not None and shs is not None) else None
        d.sh_jac = _jac_ptr(jac, N).value if jac is not None else None
        d.g_means2D = g_means2D[j].data_ptr() if g_means2D is not None else None
        d.H, d.W = st.image_height, st.image_width
    with _stage("preprocess_bwd"):
        _lib.check(lib.gs_preprocess_bwd_views(k, descs, N, K, views[0].sh_degree if k else 0, _ptr(means3D), _ptr(scales),
                                               _ptr(rotations), _ptr(cov3D_precomp), _ptr(opacities), _ptr(shs),
                                               _ptr(colors_precomp), _ptr(g_means3D), _ptr(g_scales), _ptr(g_rots),
                                               _ptr(g_cov), _ptr(g_opac), _ptr(g_shs), _ptr(g_colors),
                                               0 if accumulate_into is None else 1, _stream()),
                   "gs_preprocess_bwd_views")
    return g_means3D, g_means2D, g_shs, g_colors, g_opac, g_scales, g_rots, g_cov


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp,
                settings, t_eps, tile_rows=None, need_grad=True):
        dev = means3D.device
        with torch.cuda.device(dev):
            band = _band(settings, tile_rows)
            state = preprocess_and_bin(settings, means3D, opacities, scales, rotations, cov3D_precomp, shs,
                                       colors_precomp, tile_rows, want_jac=need_grad)
            image, depth, alpha, aux, rows = render_forward(settings, state.splat, state.ranges, state.point_list,
                                                            need_grad, t_eps, band)
        ctx.rows = rows
        ctx.set_materialize_grads(False)   # unused outputs (depth, alpha) arrive as None, not as zero fills
        ctx.settings = settings
        ctx.t_eps = float(t_eps)
        ctx.band = band
        ctx.has = tuple(t is not None for t in (shs, colors_precomp, scales, rotations, cov3D_precomp))
        ctx.means2D_shape = None if means2D is None else tuple(means2D.shape)
        saved = [means3D, opacities, state.splat, state.radii, state.point_list, state.ranges, depth, alpha]
        saved += [t for t in (shs, colors_precomp, scales, rotations, cov3D_precomp) if t is not None]
        if aux is not None:
            saved.append(aux)
        ctx.has_aux = aux is not None
        # the colour side table (N x 48 B) goes with the saved tensors: autograd frees it with them after a backward that
        # does not retain the graph, and a retained graph keeps it for the next backward
        ctx.has_jac = state.sh_jac is not None
        if ctx.has_jac:
            saved.append(state.sh_jac)
        ctx.save_for_backward(*saved)
        ctx.mark_non_differentiable(state.radii)
        return image, state.radii, depth, alpha

    @staticmethod
    def backward(ctx, g_image, g_radii, g_depth, g_alpha):
        saved = list(ctx.saved_tensors)
        means3D, opacities, splat, radii, point_list, ranges, depth, alpha = saved[:8]
        rest = saved[8:]
        opt = []
        for h in ctx.has:
            opt.append(rest.pop(0) if h else None)
        shs, colors_precomp, scales, rotations, cov3D_precomp = opt
        if not ctx.has_aux:
            raise RuntimeError("backward called but no input required grad in forward")
        aux = rest.pop(0)
        sh_jac = rest.pop(0) if ctx.has_jac else None
        settings = ctx.settings
        with torch.cuda.device(means3D.device):
            rows, ctx.rows = ctx.rows, None          # cleared by the forward; a second backward clears its own
            rows = render_backward(settings, splat, ranges, point_list, aux, depth, alpha, g_image, g_depth, g_alpha,
                                   rows, ctx.t_eps, ctx.band)
            grads = preprocess_backward(settings, means3D, opacities, scales, rotations, cov3D_precomp, shs,
                                        colors_precomp, radii, rows, ctx.needs_input_grad, ctx.means2D_shape,
                                        sh_jac=sh_jac)
        return (*grads, None, None, None, None)


def wants_grad(*tensors) -> bool:
    """Will a backward follow?  Grad mode is always off inside autograd.Function.forward, so this is asked by the
    caller, outside: under torch.no_grad() the forward then skips the backward's buffers (3 H W floats of aux, the
    clear of N 64-byte gradient rows)."""
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)


def rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp,
                        raster_settings, t_eps: float = DEFAULT_T_EPS, tile_rows=None):
    """tile_rows = (begin, end): render only that band of 16-pixel tile rows (pixels outside it are
    left unwritten) -- the unit of work of tile-row sharded inference."""
    need_grad = wants_grad(means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp)
    return _RasterizeGaussians.apply(means3D, means2D, shs, colors_precomp, opacities, scales, rotations,
                                     cov3D_precomp, raster_settings, t_eps, tile_rows, need_grad)


class GaussianRasterizer(nn.Module):
    """Drop-in for the class the reference constructs at gaussian_renderer/__init__.py:585 and calls at :635."""

    def __init__(self, raster_settings: GaussianRasterizationSettings, t_eps: float = DEFAULT_T_EPS,
                 tile_rows=None):
        super().__init__()
        self.raster_settings = raster_settings
        self.t_eps = t_eps
        self.tile_rows = tile_rows

    def forward(self, means3D, means2D=None, shs=None, colors_precomp=None, opacities=None, scales=None,
                rotations=None, cov3D_precomp=None, cov3Ds_precomp=None):
        args = check_inputs(self.raster_settings, means3D, means2D, shs, colors_precomp, opacities, scales, rotations,
                            cov3D_precomp, cov3Ds_precomp)
        return rasterize_gaussians(*args, self.raster_settings, self.t_eps, self.tile_rows)


def check_inputs(rs, means3D, means2D=None, shs=None, colors_precomp=None, opacities=None, scales=None,
                 rotations=None, cov3D_precomp=None, cov3Ds_precomp=None):
    """Argument checks of GaussianRasterizer.forward (the reference's adapter calls it by keyword,
    gaussian_renderer/__init__.py:623-638) -> the eight tensors, contiguous float32 on one device."""
    if cov3D_precomp is None:
        cov3D_precomp = cov3Ds_precomp          # the reference's own spelling (:85, :637)
    if (shs is None) == (colors_precomp is None):
        raise ValueError("Please provide exactly one of either SHs or precomputed colors!")
    if ((scales is None or rotations is None) and cov3D_precomp is None) or \
            ((scales is not None or rotations is not None) and cov3D_precomp is not None):
        raise ValueError("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
    if opacities is None:
        raise ValueError("opacities is required")
    if not torch.is_tensor(means3D) or means3D.dim() != 2 or means3D.shape[1] != 3:
        raise ValueError("means3D must be a (N,3) tensor")
    dev = means3D.device
    N = means3D.shape[0]
    if shs is not None:
        if shs.dim() != 3 or shs.shape[2] != 3:
            raise ValueError(f"shs must be (N,K,3), got {tuple(shs.shape)}")
        if (rs.sh_degree + 1) ** 2 > shs.shape[1]:
            raise ValueError(f"sh_degree {rs.sh_degree} needs {(rs.sh_degree + 1) ** 2} coefficients, shs has {shs.shape[1]}")
    means3D = _prep(means3D, "means3D", (3,), dev)
    shs = _prep(shs, "shs", None, dev)
    colors_precomp = _prep(colors_precomp, "colors_precomp", (3,), dev)
    opacities = _prep(opacities.reshape(N, 1) if opacities.dim() == 1 else opacities, "opacities", (1,), dev)
    scales = _prep(scales, "scales", (3,), dev)
    rotations = _prep(rotations, "rotations", (4,), dev)
    cov3D_precomp = _prep(cov3D_precomp, "cov3D_precomp", (6,), dev)
    for name, t in (("shs", shs), ("colors_precomp", colors_precomp), ("opacities", opacities),
                    ("scales", scales), ("rotations", rotations), ("cov3D_precomp", cov3D_precomp)):
        if t is not None and t.shape[0] != N:
            raise ValueError(f"{name} has {t.shape[0]} rows, means3D has {N}")
    if not (0 <= rs.sh_degree <= 3):
        raise ValueError(f"sh_degree must be 0..3, got {rs.sh_degree}")
    return means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp
