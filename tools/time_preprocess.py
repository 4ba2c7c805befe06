#!/usr/bin/env python3
"""Time gs_preprocess_fwd / gs_preprocess_bwd alone on the bench scene (config 3), with and without the colour side
table.  usage: python tools/time_preprocess.py [N]"""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 3_000_000
    dgr = importlib.import_module("diff_gaussian_rasterization")
    R = importlib.import_module("gaussian-splatting_amd.rasterizer")
    syn = importlib.import_module("gaussian-splatting_amd.synthetic")
    dev = torch.device("cuda:0")
    sc = syn.make_scene(N, 16, 840, 1297, seed=0)
    m3, sh, op, s, q = (getattr(sc, k).to(dev) for k in ("means3D", "shs", "opacities", "scales", "rotations"))
    st = dgr.GaussianRasterizationSettings(
        image_height=sc.H, image_width=sc.W, tanfovx=sc.tanfovx, tanfovy=sc.tanfovy, bg=sc.bg.to(dev), scale_modifier=1.0,
        viewmatrix=sc.viewmatrix.to(dev), projmatrix=sc.projmatrix.to(dev), sh_degree=sc.sh_degree,
        campos=sc.campos.to(dev), prefiltered=False, debug=False)
    i32 = dict(dtype=torch.int32, device=dev)
    out = (torch.empty((N, 12), device=dev), torch.empty((N,), **i32), torch.empty((N,), **i32), torch.empty((N, 2), **i32))
    jac = torch.empty((N, R.sh_jac_row_floats()), device=dev)
    rows = torch.randn((N, R.grad_row_floats()), device=dev)
    _, radii, *_ = R.preprocess(st, m3, op, s, q, None, sh, None, out=out, sh_jac=jac)
    needs = [True, True, True, False, True, True, True, False]
    g = R.preprocess_backward(st, m3, op, s, q, None, sh, None, radii, rows, needs, (N, 3))
    print(f"N = {N}, drawn {(radii > 0).sum().item()}")
    print(f"preprocess_fwd  no table : {timed(lambda: R.preprocess(st, m3, op, s, q, None, sh, None, out=out)):.4f} ms")
    print(f"preprocess_fwd  + table  : {timed(lambda: R.preprocess(st, m3, op, s, q, None, sh, None, out=out, sh_jac=jac)):.4f} ms")
    print(f"preprocess_bwd  two launches (re-reads shs): "
          f"{timed(lambda: R.preprocess_backward(st, m3, op, s, q, None, sh, None, radii, rows, needs, (N, 3), out=g)):.4f} ms")
    print(f"preprocess_bwd  one launch (table)         : "
          f"{timed(lambda: R.preprocess_backward(st, m3, op, s, q, None, sh, None, radii, rows, needs, (N, 3), out=g, sh_jac=jac)):.4f} ms")


main()
