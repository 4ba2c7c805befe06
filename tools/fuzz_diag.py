#!/usr/bin/env python3
"""Where does a failing fuzz case differ?  usage: python tools/fuzz_diag.py seed [seed ...]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fuzz_parity as F  # noqa: E402


def main():
    import diff_gaussian_rasterization as dgr
    from oracle import oracle, oracle64
    oracle.build()
    oracle64.build()
    dev = torch.device("cuda:0")
    for seed in map(int, sys.argv[1:]):
        c = F.make_case(seed)
        outs, grads, geom = oracle.forward_backward(
            c["H"], c["W"], c["deg"], c["viewmatrix"], c["projmatrix"], c["campos"], c["bg"], c["means3D"],
            c["opacities"], scales=c["scales"], rotations=c["rotations"], cov3D_precomp=c["cov3D_precomp"],
            shs=c["shs"], colors_precomp=c["colors_precomp"], G_C=c["G_C"], G_Z=c["G_Z"], G_A=c["G_A"])
        _, grads64, geom64 = oracle64.forward_backward(
            c["H"], c["W"], c["deg"], c["viewmatrix"], c["projmatrix"], c["campos"], c["bg"], c["means3D"],
            c["opacities"], scales=c["scales"], rotations=c["rotations"], cov3D_precomp=c["cov3D_precomp"],
            shs=c["shs"], colors_precomp=c["colors_precomp"], G_C=c["G_C"], G_Z=c["G_Z"], G_A=c["G_A"])
        t = lambda a: torch.tensor(a, device=dev)  # noqa: E731
        inp = {k: (None if c[k] is None else t(c[k]).requires_grad_(True)) for k in F.INPUTS}
        s = dgr.GaussianRasterizationSettings(
            image_height=c["H"], image_width=c["W"], tanfovx=c["tanfovx"], tanfovy=c["tanfovy"], bg=t(c["bg"]),
            scale_modifier=1.0, viewmatrix=t(c["viewmatrix"]), projmatrix=t(c["projmatrix"]), sh_degree=c["deg"],
            campos=t(c["campos"]), prefiltered=False, debug=False)
        m2 = torch.zeros((c["N"], 3), device=dev, requires_grad=True)
        image, radii, depth, alpha = dgr.GaussianRasterizer(s, t_eps=0.0)(
            means3D=inp["means3D"], means2D=m2, shs=inp["shs"], colors_precomp=inp["colors_precomp"],
            opacities=inp["opacities"], scales=inp["scales"], rotations=inp["rotations"],
            cov3D_precomp=inp["cov3D_precomp"])
        ((image * t(c["G_C"])).sum() + (depth * t(c["G_Z"])).sum() + (alpha * t(c["G_A"])).sum()).backward()
        print(f"== seed {seed}: N={c['N']} {c['W']}x{c['H']} drawn {(geom.radii > 0).sum()} pairs {geom.tiles_touched.sum()}")
        print("   image err", np.abs(image.detach().cpu().numpy() - outs["render"]).max(),
              "alpha err", np.abs(alpha.detach().cpu().numpy() - outs["alpha"]).max())
        print("   boxes of the float64 evaluation equal:", np.array_equal(geom.rect, geom64.rect))
        ms = m2.grad.cpu().numpy()[:, :2] / np.array([0.5 * c["W"], 0.5 * c["H"]], np.float32)
        e32, e64 = np.abs(ms - grads["means_screen"]).max(), np.abs(ms - grads64["means_screen"]).max()
        print(f"   d_means_screen: max|ref| {np.abs(grads64['means_screen']).max():.3e} hip-oracle32 {e32:.3e} hip-oracle64 {e64:.3e} "
              f"oracle32-oracle64 {np.abs(grads['means_screen'] - grads64['means_screen']).max():.3e}")
        for k in F.INPUTS:
            if inp[k] is None:
                continue
            got, ref = inp[k].grad.cpu().numpy().reshape(c["N"], -1), grads[k].reshape(c["N"], -1)
            ref64 = grads64[k].reshape(c["N"], -1)
            err = np.abs(got - ref)
            print(f"   d_{k}: max|ref| {np.abs(ref).max():.3e} max err {err.max():.3e} | vs float64: hip {np.abs(got - ref64).max():.3e} "
                  f"oracle32 {np.abs(ref - ref64).max():.3e}")
            for i in np.argsort(-err.max(axis=1))[:3]:
                j = int(err[i].argmax())
                print(f"      gaussian {i} comp {j}: hip {got[i, j]:+.6e} oracle {ref[i, j]:+.6e} f64 {ref64[i, j]:+.6e} | radius {geom.radii[i]} "
                      f"depth {geom.depth[i]:.3f} opacity {c['opacities'][i, 0]:.4f} conic {geom.conic[i]} rect {geom.rect[i]}")


main()
