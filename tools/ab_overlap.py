"""A/B of parallel.OVERLAP_BINNING on the 8-view step (config 4: 6 M Gaussians; or N from argv): ms per step, forward and backward apart.
usage: python tools/ab_overlap.py [N]"""
import importlib, os, sys, torch
sys.path.insert(0, os.getcwd())
par = importlib.import_module("gaussian-splatting_amd.parallel")
syn = importlib.import_module("gaussian-splatting_amd.synthetic")
dgr = importlib.import_module("diff_gaussian_rasterization")
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 6_000_000
sc = syn.make_scene(N, 16, 840, 1297, seed=0)
def settings(v):
    return dgr.GaussianRasterizationSettings(image_height=v.H, image_width=v.W, tanfovx=v.tanfovx, tanfovy=v.tanfovy, bg=v.bg.to(dev), scale_modifier=1.0, viewmatrix=v.viewmatrix.to(dev), projmatrix=v.projmatrix.to(dev), sh_degree=v.sh_degree, campos=v.campos.to(dev), prefiltered=False, debug=False)
sets = [settings(syn.jitter_view(sc, v)) for v in range(8)]
p = {k: getattr(sc, k).to(dev).requires_grad_(True) for k in ("means3D", "shs", "opacities", "scales", "rotations")}
G = sc.G.to(dev)
step = par.ViewShardedStep(sets)
def run(n):
    tf = tb = 0.0
    for _ in range(n):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record()
        images, _, _, _ = step(means3D=p["means3D"], shs=p["shs"], opacities=p["opacities"], scales=p["scales"], rotations=p["rotations"])
        e[1].record()
        sum((im * G).sum() for im in images).backward()
        e[2].record(); torch.cuda.synchronize()
        for t in p.values(): t.grad = None
        tf += e[0].elapsed_time(e[1]); tb += e[1].elapsed_time(e[2])
    return tf / n, tb / n
for rep in range(2):
    for flag in (False, True):
        par.OVERLAP_BINNING = flag
        run(2)
        f, b = run(5)
        print(f"overlap={flag}: forward {f:.3f} ms, backward {b:.3f} ms, step {f + b:.3f} ms")
