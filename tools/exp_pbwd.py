"""Experiment: preprocess_bwd time against which gradients are requested (bytes written)."""
import importlib, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diff_gaussian_rasterization as dgr
syn = importlib.import_module("gaussian-splatting_amd.synthetic")
prof = importlib.import_module("gaussian-splatting_amd.profiling")
N, K, H, W = 3_000_000, 16, 840, 1297
sc = syn.make_scene(N, K, H, W, seed=0)
dev = torch.device("cuda:0")
G = sc.G.to(dev)
st = dgr.GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=sc.tanfovx, tanfovy=sc.tanfovy, bg=sc.bg.to(dev),
    scale_modifier=1.0, viewmatrix=sc.viewmatrix.to(dev), projmatrix=sc.projmatrix.to(dev), sh_degree=3, campos=sc.campos.to(dev),
    prefiltered=False, debug=False)
rast = dgr.GaussianRasterizer(st)
names = ("means3D", "shs", "opacities", "scales", "rotations")
for req in (names, ("means3D", "opacities", "scales", "rotations"), ("shs",), ("opacities",)):
    p = {k: getattr(sc, k).to(dev).requires_grad_(k in req) for k in names}
    for it in range(4):
        if it == 1:
            prof.enable()
        for v in p.values():
            v.grad = None
        img, radii, d, a = rast(means3D=p["means3D"], means2D=None, shs=p["shs"], opacities=p["opacities"], scales=p["scales"], rotations=p["rotations"])
        img.backward(G)
    ms = prof.disable_and_collect()
    print(req, {k: round(v, 4) for k, v in ms.items() if "preprocess" in k}, flush=True)
