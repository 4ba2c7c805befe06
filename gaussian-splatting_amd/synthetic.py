"""Synthetic scene generator for benchmarks, goldens and tests.

This is the generator SURVEY.md section 8d specifies: a pinhole camera at the
origin looking down +z, N Gaussians with ~1.8 px median screen sigma spread
over the frustum (10 % margin outside), SH coefficients of K = (deg+1)^2.
Matrices are in the column-vector convention the reference rasterizer applies
them in (cam = V @ [p;1], clip = P @ [p;1];
reference gaussian_renderer/__init__.py:122,125).

All draws come from one CPU ``torch.Generator`` in a fixed order so that a
(seed, N, K, H, W) tuple names the same scene on every machine.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Optional

import torch


@dataclass
class Scene:
    H: int
    W: int
    sh_degree: int
    means3D: torch.Tensor      # (N,3)
    scales: torch.Tensor       # (N,3)
    rotations: torch.Tensor    # (N,4) normalised (r,x,y,z)
    opacities: torch.Tensor    # (N,1)
    shs: torch.Tensor          # (N,K,3)
    viewmatrix: torch.Tensor   # (4,4)
    projmatrix: torch.Tensor   # (4,4) full projection P @ V
    campos: torch.Tensor       # (3,)
    bg: torch.Tensor           # (3,)
    tanfovx: float
    tanfovy: float
    G: Optional[torch.Tensor] = None   # (3,H,W) upstream gradient of render

    def to(self, device):
        out = Scene(**self.__dict__)
        for k, v in self.__dict__.items():
            if isinstance(v, torch.Tensor):
                setattr(out, k, v.to(device))
        return out


def pinhole(H: int, W: int, fovx_deg: float = 60.0, znear: float = 0.01, zfar: float = 100.0):
    """Projection matrix of SURVEY.md 8d (column-vector convention, w = z)."""
    tanx = math.tan(math.radians(fovx_deg) * 0.5)
    tany = tanx * H / W
    P = torch.zeros(4, 4, dtype=torch.float32)
    P[0, 0] = 1.0 / tanx
    P[1, 1] = 1.0 / tany
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -zfar * znear / (zfar - znear)
    P[3, 2] = 1.0
    return P, tanx, tany


def make_scene(N: int, K: int, H: int, W: int, seed: int = 0, sh_degree: Optional[int] = None,
               with_G: bool = True) -> Scene:
    g = torch.Generator().manual_seed(seed)
    P, tanx, tany = pinhole(H, W)
    fx = W / (2.0 * tanx)
    z = torch.rand(N, generator=g) * 8.0 + 2.0
    x = (torch.rand(N, generator=g) * 2.2 - 1.1) * tanx * z
    y = (torch.rand(N, generator=g) * 2.2 - 1.1) * tany * z
    scales = torch.exp(torch.randn(N, 3, generator=g) * 0.5) * (1.5 * z / fx)[:, None]
    q = torch.randn(N, 4, generator=g)
    q = q / q.norm(dim=1, keepdim=True)
    opacity = torch.rand(N, 1, generator=g) * 0.98 + 0.01
    sh = torch.randn(N, K, 3, generator=g) * 0.2
    sh[:, 0, :] = torch.rand(N, 3, generator=g) * 3.0 - 1.5
    V = torch.eye(4, dtype=torch.float32)
    G = (torch.rand(3, H, W, generator=g) * 2.0 - 1.0) if with_G else None
    if sh_degree is None:
        sh_degree = int(round(math.sqrt(K))) - 1
    return Scene(H=H, W=W, sh_degree=sh_degree,
                 means3D=torch.stack([x, y, z], dim=1).contiguous(),
                 scales=scales.contiguous(), rotations=q.contiguous(),
                 opacities=opacity.contiguous(), shs=sh.contiguous(),
                 viewmatrix=V, projmatrix=(P @ V).contiguous(),
                 campos=torch.zeros(3), bg=torch.zeros(3),
                 tanfovx=tanx, tanfovy=tany, G=G)


def jitter_view(scene: Scene, view_seed: int, sigma: float = 0.2) -> Scene:
    """Same Gaussians, camera translated by sigma*randn(3).

    viewmatrix = [I | -c] so the camera centre is c; projmatrix = P @ viewmatrix.
    """
    g = torch.Generator().manual_seed(1000 + view_seed)
    c = torch.randn(3, generator=g) * sigma
    P, _, _ = pinhole(scene.H, scene.W)
    V = torch.eye(4, dtype=torch.float32)
    V[:3, 3] = -c
    out = Scene(**scene.__dict__)
    dev = scene.means3D.device
    out.viewmatrix = V.to(dev)
    out.projmatrix = (P @ V).contiguous().to(dev)
    out.campos = c.to(dev)
    return out
