#!/usr/bin/env python3
"""Counters of the one-wave-per-tile backward on the bench scene (config 3): batches, wave-visits, block-visits with a
live lane, slots, flush iterations.  Needs a library built with -DGS_BWD4_STATS (see tools/README.md):
GSPLAT_LIB=gaussian-splatting_amd/lib/libgsplat_stats.so python tools/bwd4_stats.py [t_eps]"""
import ctypes as C
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    t_eps = float(sys.argv[1]) if len(sys.argv) > 1 else 1e-5
    N, K, H, W = 3_000_000, 16, 840, 1297
    if len(sys.argv) > 2:
        N, H, W = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    import diff_gaussian_rasterization as dgr
    syn = importlib.import_module("gaussian-splatting_amd.synthetic")
    lib = importlib.import_module("gaussian-splatting_amd._lib").load()
    dev = torch.device("cuda:0")
    sc = syn.make_scene(N, K, H, W, seed=0).to(dev)
    st = dgr.GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=sc.tanfovx, tanfovy=sc.tanfovy, bg=sc.bg,
                                           scale_modifier=1.0, viewmatrix=sc.viewmatrix, projmatrix=sc.projmatrix,
                                           sh_degree=sc.sh_degree, campos=sc.campos, prefiltered=False, debug=False)
    rast = dgr.GaussianRasterizer(st, t_eps=t_eps)
    inp = dict(means3D=sc.means3D, shs=sc.shs, opacities=sc.opacities, scales=sc.scales, rotations=sc.rotations)
    inp = {k: v.clone().requires_grad_(True) for k, v in inp.items()}
    buf = (C.c_ulonglong * 8)()
    for it in range(2):
        image, radii, depth, alpha = rast(means3D=inp["means3D"], means2D=None, shs=inp["shs"], opacities=inp["opacities"],
                                          scales=inp["scales"], rotations=inp["rotations"])
        torch.cuda.synchronize()
        lib.gs_debug_bwd4_stats.argtypes = [C.c_void_p, C.c_int]
        lib.gs_debug_bwd4_stats(buf, 1)
        (image * sc.G).sum().backward()
        torch.cuda.synchronize()
        lib.gs_debug_bwd4_stats(buf, 1)
    names = ["batches", "wave_visits", "live_block_visits", "sum_n_eff", "sum_slots", "flush_iters", "live_lane_pixels"]
    d = dict(zip(names, list(buf)))
    print(d)
    print("visits/batch %.2f  live blocks/visit %.2f of 16  pairs/batch %.2f  slots/pair %.2f  flush iters/batch %.2f  live px/visit %.1f of 256" % (
        d["wave_visits"] / d["batches"], d["live_block_visits"] / d["wave_visits"], d["sum_n_eff"] / d["batches"],
        d["sum_slots"] / d["sum_n_eff"], d["flush_iters"] / d["batches"], d["live_lane_pixels"] / d["wave_visits"]))


main()
