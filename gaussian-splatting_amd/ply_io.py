"""Trained-model PLY <-> GPU tensors: host side of gs_ply_parse_header / gs_ply_unpack /
gs_ply_write_header / gs_ply_pack.

Mirrors GaussianModel.load_ply / save_ply (reference scene/gaussian_model.py:377-459, :333-375): the same file,
the same six parameter tensors (xyz (N,3), features_dc (N,1,3), features_rest (N,K-1,3), opacity (N,1),
scaling (N,3), rotation (N,4)).  The header is parsed in the C library on the host; the vertex records go to
the GPU as they are and are unpacked there (no per-attribute numpy copies, no float64 detour, no plyfile).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict

import numpy as np
import torch

from . import _lib

RAW = ("xyz", "features_dc", "features_rest", "opacity", "scaling", "rotation")


def parse_header(head: bytes, file_size: int = None) -> _lib.GsPlyLayout:
    """head: the first bytes of the file (the header must end inside them); file_size defaults to len(head)."""
    lib = _lib.load()
    layout = _lib.GsPlyLayout()
    head = bytes(head)
    _lib.check(lib.gs_ply_parse_header(head, len(head), len(head) if file_size is None else int(file_size),
                                       C.byref(layout)), "gs_ply_parse_header")
    return layout


def load_ply(path: str, device="cuda", activated: bool = False) -> Dict[str, torch.Tensor]:
    """-> {"xyz", "features_dc", "features_rest", "opacity", "scaling", "rotation", "sh_degree"} on `device`;
    with activated=True also the rasterizer's inputs {"shs", "opacities", "scales", "rotations"}
    (reference :104-142: cat / sigmoid / exp / normalize)."""
    lib = _lib.load()
    dev = torch.device(device)
    if dev.type != "cuda":
        raise RuntimeError("load_ply unpacks on the GPU; there is no CPU path")
    size = os.path.getsize(path)
    with open(path, "rb") as f:
        head = f.read(min(size, 1 << 16))
    layout = parse_header(head, size)
    N, stride, off = int(layout.num_vertices), int(layout.stride_bytes), int(layout.data_offset)
    R = int(layout.sh_rest_count)
    with torch.cuda.device(dev):
        if N:
            staging = torch.empty((N * stride,), dtype=torch.uint8, pin_memory=True)   # file -> pinned -> HBM, one pass
            with open(path, "rb") as f:
                f.seek(off)
                if f.readinto(staging.numpy()) != N * stride:
                    raise RuntimeError(f"{path}: short read")
            records = staging.to(dev, non_blocking=True)
        else:
            records = torch.empty((0,), dtype=torch.uint8, device=dev)
        f32 = dict(dtype=torch.float32, device=dev)
        out = {"xyz": torch.empty((N, 3), **f32), "features_dc": torch.empty((N, 1, 3), **f32),
               "features_rest": torch.empty((N, R // 3, 3), **f32), "opacity": torch.empty((N, 1), **f32),
               "scaling": torch.empty((N, 3), **f32), "rotation": torch.empty((N, 4), **f32)}
        act = {}
        if activated:
            act = {"shs": torch.empty((N, R // 3 + 1, 3), **f32), "opacities": torch.empty((N, 1), **f32),
                   "scales": torch.empty((N, 3), **f32), "rotations": torch.empty((N, 4), **f32)}
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None and t.numel() else None  # noqa: E731
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        _lib.check(lib.gs_ply_unpack(C.byref(layout), p(records), *[p(out[k]) for k in RAW],
                                     p(act.get("shs")), p(act.get("opacities")), p(act.get("scales")),
                                     p(act.get("rotations")), st), "gs_ply_unpack")
    out.update(act)
    out["sh_degree"] = int(round((R // 3 + 1) ** 0.5)) - 1
    return out


def save_ply(path: str, xyz, features_dc, features_rest, opacity, scaling, rotation):
    """Writes the file GaussianModel.save_ply writes (reference :333-375) from GPU tensors."""
    lib = _lib.load()
    dev = xyz.device
    if dev.type != "cuda":
        raise RuntimeError("save_ply packs on the GPU; there is no CPU path")
    N = xyz.shape[0]
    t = [x.detach().to(torch.float32).contiguous() for x in (xyz, features_dc, features_rest, opacity, scaling, rotation)]
    R = t[2].numel() // N if N else int(np.prod(features_rest.shape[1:]))
    if t[1].numel() != 3 * N or t[3].numel() != N or t[4].numel() != 3 * N or t[5].numel() != 4 * N or R % 3:
        raise ValueError("save_ply: parameter shapes do not match (N,1,3) (N,K-1,3) (N,1) (N,3) (N,4)")
    layout = _lib.GsPlyLayout()
    n = lib.gs_ply_write_header(None, 0, N, R, C.byref(layout))
    buf = C.create_string_buffer(int(n))
    lib.gs_ply_write_header(buf, n, N, R, C.byref(layout))
    with torch.cuda.device(dev):
        records = torch.empty((N * layout.stride_bytes,), dtype=torch.uint8, device=dev)
        p = lambda x: C.c_void_p(x.data_ptr()) if x.numel() else None  # noqa: E731
        _lib.check(lib.gs_ply_pack(C.byref(layout), p(records), *[p(x) for x in t],
                                   C.c_void_p(torch.cuda.current_stream().cuda_stream)), "gs_ply_pack")
        host = records.cpu().numpy()
    d = os.path.dirname(path)
    if d:
        os.makedirs(d, exist_ok=True)
    with open(path, "wb") as f:
        f.write(buf.raw[:int(n)])
        f.write(host.tobytes())
