"""ctypes wrapper around oracle/libgs_oracle.so (the CPU parity oracle).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never by the product package.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# oracle64.py runs this file a second time with GSO_DOUBLE set: the float64 build of the same C source
_DOUBLE = bool(globals().get("GSO_DOUBLE", False))
REAL = np.float64 if _DOUBLE else np.float32
_LIB_NAME = "libgs_oracle_f64.so" if _DOUBLE else "libgs_oracle.so"
_LIB_PATH = os.path.join(_HERE, _LIB_NAME)
_lib = None

TILE = 16


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "gs_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", _LIB_NAME], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.gso_scan.restype = C.c_int64
        _lib.gso_set_t_eps.argtypes = [C.c_double]
        _lib.gso_get_t_eps.restype = C.c_double
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=REAL)


def set_threads(n: int):
    lib().gso_set_threads(int(n))


def max_threads() -> int:
    return int(lib().gso_max_threads())


class Geom:
    """Per-Gaussian outputs of gso_preprocess."""
    pass


def preprocess(H, W, sh_degree, viewmatrix, projmatrix, campos, means3D, scales=None, rotations=None,
               cov3D_precomp=None, shs=None, colors_precomp=None) -> Geom:
    means3D = _f32(means3D)
    N = means3D.shape[0]
    scales, rotations, cov3D_precomp = _f32(scales), _f32(rotations), _f32(cov3D_precomp)
    shs, colors_precomp = _f32(shs), _f32(colors_precomp)
    K = shs.shape[1] if shs is not None else 0
    g = Geom()
    g.N, g.K, g.H, g.W, g.deg = N, K, int(H), int(W), int(sh_degree)
    g.view, g.proj, g.campos = _f32(viewmatrix).reshape(16), _f32(projmatrix).reshape(16), _f32(campos)
    g.means3D, g.scales, g.rotations, g.cov6, g.shs, g.colors_precomp = (
        means3D, scales, rotations, cov3D_precomp, shs, colors_precomp)
    g.xy = np.zeros((N, 2), REAL)
    g.depth = np.zeros(N, REAL)
    g.cov2d = np.zeros((N, 3), REAL)
    g.conic = np.zeros((N, 3), REAL)
    g.rgb = np.zeros((N, 3), REAL)
    g.rect = np.zeros((N, 4), np.int32)
    g.radii = np.zeros(N, np.int32)
    g.tiles_touched = np.zeros(N, np.uint32)
    g.clampmask = np.zeros(N, np.uint8)
    rc = lib().gso_preprocess(N, K, g.deg, g.H, g.W, _p(g.view), _p(g.proj), _p(g.campos), _p(means3D),
                              _p(scales), _p(rotations), _p(cov3D_precomp), _p(shs), _p(colors_precomp),
                              _p(g.xy), _p(g.depth), _p(g.cov2d), _p(g.conic), _p(g.rgb), _p(g.rect),
                              _p(g.radii), _p(g.tiles_touched), _p(g.clampmask))
    if rc != 0:
        raise ValueError(f"gso_preprocess failed: {rc}")
    return g


def binning(g: Geom):
    """keys/values (sorted), offsets, D, per-tile ranges -- the integer stage."""
    offsets = np.zeros(g.N, np.uint32)
    D = int(lib().gso_scan(g.N, _p(g.tiles_touched), _p(offsets)))
    keys = np.zeros(max(D, 1), np.uint64)
    vals = np.zeros(max(D, 1), np.uint32)
    lib().gso_duplicate_with_keys(g.N, g.W, _p(g.rect), _p(g.depth), _p(offsets), _p(g.tiles_touched),
                                  _p(keys), _p(vals))
    unsorted = (keys[:D].copy(), vals[:D].copy())
    tiles_x, tiles_y = (g.W + TILE - 1) // TILE, (g.H + TILE - 1) // TILE
    tile_bits = max(1, int(tiles_x * tiles_y - 1).bit_length())
    rc = lib().gso_sort_pairs(C.c_int64(D), 32 + tile_bits, _p(keys), _p(vals))
    assert rc == 0
    ranges = np.zeros((tiles_x * tiles_y, 2), np.uint32)
    rc = lib().gso_tile_ranges(C.c_int64(D), tiles_x * tiles_y, _p(keys), _p(ranges))
    assert rc == 0
    return dict(D=D, offsets=offsets, keys=keys[:D], vals=vals[:D], ranges=ranges,
                unsorted_keys=unsorted[0], unsorted_vals=unsorted[1])


def _set_t_eps(t_eps):
    """Stop threshold of the HIP path's switchable deviation (gso_set_t_eps); 0 = the reference, which never stops."""
    lib().gso_set_t_eps(float(t_eps))


def render_forward(g: Geom, opacities, bg, t_eps=0.0):
    _set_t_eps(t_eps)
    op = _f32(opacities).reshape(-1)
    bg = _f32(bg)
    img = np.zeros((3, g.H, g.W), REAL)
    dep = np.zeros((1, g.H, g.W), REAL)
    alp = np.zeros((1, g.H, g.W), REAL)
    rc = lib().gso_render_forward(g.N, g.H, g.W, _p(bg), _p(g.xy), _p(g.depth), _p(g.conic), _p(g.rgb),
                                  _p(op), _p(g.rect), _p(g.tiles_touched), _p(img), _p(dep), _p(alp))
    assert rc == 0
    return img, dep, alp


def render_backward(g: Geom, opacities, G_C, G_Z=None, G_A=None, t_eps=0.0):
    _set_t_eps(t_eps)
    op = _f32(opacities).reshape(-1)
    G_C = _f32(G_C)
    G_Z = np.zeros((1, g.H, g.W), REAL) if G_Z is None else _f32(G_Z)
    G_A = np.zeros((1, g.H, g.W), REAL) if G_A is None else _f32(G_A)
    out = dict(d_xy=np.zeros((g.N, 2), REAL), d_conic=np.zeros((g.N, 3), REAL),
               d_rgb=np.zeros((g.N, 3), REAL), d_opacity=np.zeros((g.N, 1), REAL),
               d_depth=np.zeros(g.N, REAL))
    rc = lib().gso_render_backward(g.N, g.H, g.W, _p(g.xy), _p(g.depth), _p(g.conic), _p(g.rgb), _p(op),
                                   _p(g.rect), _p(g.tiles_touched), _p(G_C), _p(G_Z), _p(G_A),
                                   _p(out["d_xy"]), _p(out["d_conic"]), _p(out["d_rgb"]),
                                   _p(out["d_opacity"]), _p(out["d_depth"]))
    assert rc == 0
    return out


def preprocess_backward(g: Geom, rb: dict):
    N, K = g.N, g.K
    res = dict(means3D=np.zeros((N, 3), REAL))
    res["scales"] = np.zeros((N, 3), REAL) if g.scales is not None and g.cov6 is None else None
    res["rotations"] = np.zeros((N, 4), REAL) if g.rotations is not None and g.cov6 is None else None
    res["cov3D_precomp"] = np.zeros((N, 6), REAL) if g.cov6 is not None else None
    res["shs"] = np.zeros((N, K, 3), REAL) if g.shs is not None else None
    res["colors_precomp"] = np.zeros((N, 3), REAL) if g.colors_precomp is not None else None
    rc = lib().gso_preprocess_backward(
        N, K, g.deg, g.H, g.W, _p(g.view), _p(g.proj), _p(g.campos), _p(g.means3D), _p(g.scales),
        _p(g.rotations), _p(g.cov6), _p(g.shs), _p(g.tiles_touched), _p(g.clampmask), _p(g.cov2d),
        _p(rb["d_xy"]), _p(rb["d_conic"]), _p(rb["d_rgb"]), _p(rb["d_depth"]),
        _p(res["means3D"]), _p(res["scales"]), _p(res["rotations"]), _p(res["cov3D_precomp"]),
        _p(res["shs"]), _p(res["colors_precomp"]))
    assert rc == 0
    res["opacities"] = rb["d_opacity"]
    res["means_screen"] = rb["d_xy"]
    return res


def forward_backward(H, W, sh_degree, viewmatrix, projmatrix, campos, bg, means3D, opacities, scales=None,
                     rotations=None, cov3D_precomp=None, shs=None, colors_precomp=None, G_C=None,
                     G_Z=None, G_A=None, t_eps=0.0):
    """Whole pipeline; returns (outputs dict, grads dict or None, Geom).  t_eps > 0: the HIP path's stop threshold
    (not in the reference; 0 restates it)."""
    g = preprocess(H, W, sh_degree, viewmatrix, projmatrix, campos, means3D, scales, rotations,
                   cov3D_precomp, shs, colors_precomp)
    img, dep, alp = render_forward(g, opacities, bg, t_eps)
    outs = dict(render=img, depth=dep, alpha=alp, radii=g.radii)
    grads = None
    if G_C is not None:
        rb = render_backward(g, opacities, G_C, G_Z, G_A, t_eps)
        grads = preprocess_backward(g, rb)
        grads["render_bwd"] = rb
    return outs, grads, g
