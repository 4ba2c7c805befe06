"""Nearest-neighbour distances for the scale initialisation: host side of gs_nearest_dist2.

``distCUDA2(points)`` has the name, argument and result of the reference's function
(scene/gaussian_model.py:729-746; called at :213-216): (N,3) float points on the GPU -> (N,) squared distance to
the nearest other point.  The reference builds the N x N distance matrix (out of memory at 100 k points); this
is a uniform-grid search in O(N) memory, exact.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib


def distCUDA2(points: torch.Tensor, return_stats: bool = False):
    lib = _lib.load()
    if not points.is_cuda:
        raise RuntimeError("distCUDA2 needs points on the GPU (no CPU path)")
    if points.dim() != 2 or points.shape[1] != 3:
        raise ValueError(f"points must be (N,3), got {tuple(points.shape)}")
    pts = points.detach().to(torch.float32).contiguous()
    N = pts.shape[0]
    dev = pts.device
    with torch.cuda.device(dev):
        out = torch.empty((N,), dtype=torch.float32, device=dev)
        nbytes = lib.gs_nearest_dist2_workspace_bytes(N)
        ws = torch.empty((nbytes + 256,), dtype=torch.uint8, device=dev)
        shift = (-ws.data_ptr()) % 256
        stats = (C.c_int * 5)()
        _lib.check(lib.gs_nearest_dist2(N, C.c_void_p(pts.data_ptr()) if N else None,
                                        C.c_void_p(out.data_ptr()) if N else None,
                                        C.c_void_p(ws.data_ptr() + shift), nbytes, stats if return_stats else None,
                                        C.c_void_p(torch.cuda.current_stream().cuda_stream)), "gs_nearest_dist2")
    if return_stats:
        return out, {"grid": (stats[0], stats[1], stats[2]), "isolated": stats[3], "crowded": stats[4]}
    return out
