#!/usr/bin/env python3
"""Frame time and parity margins of a full-size configuration as a function of the stop threshold t_eps: the HIP path at each
threshold against the never-stopping oracle (the reference's semantics), errors as fractions of the stated bars
(image 1e-4, depth 1e-3, gradients 1e-3 max|grad|).  usage: t_eps_margins.py [N H W] [t_eps ...]   (test infrastructure)"""
import importlib, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle
import diff_gaussian_rasterization as dgr
syn = importlib.import_module("gaussian-splatting_amd.synthetic")
args = sys.argv[1:]
N, H, W = (int(args.pop(0)), int(args.pop(0)), int(args.pop(0))) if len(args) >= 3 and "." not in args[0] and "e" not in args[0] else (3_000_000, 840, 1297)
ts = [float(a) for a in args] or [0.0, 1e-5, 2e-5, 5e-5, 1e-4]
oracle.build(); oracle.set_threads(min(oracle.max_threads(), 16))
sc = syn.make_scene(N, 16, H, W, seed=0)
a = lambda t: t.numpy()
outs, grads, geom = oracle.forward_backward(H, W, sc.sh_degree, a(sc.viewmatrix), a(sc.projmatrix), a(sc.campos), a(sc.bg), a(sc.means3D), a(sc.opacities), scales=a(sc.scales), rotations=a(sc.rotations), shs=a(sc.shs), G_C=a(sc.G))
dev = torch.device("cuda:0")
names = ("means3D", "shs", "opacities", "scales", "rotations")
p = {k: getattr(sc, k).to(dev).requires_grad_(True) for k in names}
st = dgr.GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=sc.tanfovx, tanfovy=sc.tanfovy, bg=sc.bg.to(dev), scale_modifier=1.0, viewmatrix=sc.viewmatrix.to(dev), projmatrix=sc.projmatrix.to(dev), sh_degree=sc.sh_degree, campos=sc.campos.to(dev), prefiltered=False, debug=False)
G = sc.G.to(dev)
for t in ts:
    r = dgr.GaussianRasterizer(st, t_eps=t)
    def step():
        for v in p.values(): v.grad = None
        img, radii, dep, alp = r(means3D=p["means3D"], means2D=None, shs=p["shs"], opacities=p["opacities"], scales=p["scales"], rotations=p["rotations"])
        img.backward(G); return img, dep
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): img, dep = step()
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 20 * 1e3
    e_img = np.abs(img.detach().cpu().numpy() - outs["render"]).max() / 1e-4
    e_dep = np.abs(dep.detach().cpu().numpy() - outs["depth"]).max() / 1e-3
    eg = {k: float(np.abs(p[k].grad.cpu().numpy() - grads[k]).max() / (1e-3 * np.abs(grads[k]).max())) for k in names}
    print(f"t_eps {t:g}: {ms:.3f} ms/frame; of the bars: image {e_img:.3f} depth {e_dep:.3f} " + " ".join(f"{k} {v:.3f}" for k, v in eg.items()), flush=True)
