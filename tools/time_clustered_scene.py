"""Forward+backward time of the rasterizer on an uneven scene (half of the Gaussians pulled into three regions):
what a real capture does to the tile lists next to the evenly loaded SURVEY 8d scene."""
import importlib, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
impl = importlib.import_module("gaussian-splatting_amd")
syn = importlib.import_module("gaussian-splatting_amd.synthetic")
prof = importlib.import_module("gaussian-splatting_amd.profiling")
dev = torch.device("cuda:0")
N, K, H, W = 3_000_000, 16, 840, 1297
sc = syn.make_scene(N, K, H, W, seed=0)
for frac in (0.0, 0.5, 0.8):
    g0 = torch.Generator().manual_seed(1)
    m = torch.rand(N, generator=g0) < frac
    centres = torch.tensor([[0.6, -0.3, 4.0], [-1.5, 0.5, 6.0], [1.0, 0.8, 3.0]])
    pick = centres[torch.randint(0, 3, (N,), generator=g0)]
    means = torch.where(m[:, None], pick + 0.25 * torch.randn(N, 3, generator=g0), sc.means3D).contiguous()
    p = {"means3D": means, "shs": sc.shs, "opacities": sc.opacities, "scales": sc.scales, "rotations": sc.rotations}
    p = {k: v.to(dev).requires_grad_(True) for k, v in p.items()}
    st = impl.GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=sc.tanfovx, tanfovy=sc.tanfovy, bg=sc.bg.to(dev),
                                            scale_modifier=1.0, viewmatrix=sc.viewmatrix.to(dev), projmatrix=sc.projmatrix.to(dev),
                                            sh_degree=sc.sh_degree, campos=sc.campos.to(dev), prefiltered=False, debug=False)
    rast = impl.GaussianRasterizer(st)
    G = sc.G.to(dev)
    def step():
        for t in p.values():
            t.grad = None
        rast(means3D=p["means3D"], shs=p["shs"], opacities=p["opacities"], scales=p["scales"], rotations=p["rotations"])[0].backward(G)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 10 * 1e3
    prof.enable()
    step()
    stages = prof.disable_and_collect()
    with torch.no_grad():
        b = impl.preprocess_and_bin(st, p["means3D"].detach(), p["opacities"].detach(), p["scales"].detach(), p["rotations"].detach(), None, p["shs"].detach(), None)
    ln = (b.ranges[:, 1] - b.ranges[:, 0]).long()
    print(f"clustered {frac}: {ms:.2f} ms/frame  pairs {b.num_pairs}  tile max {int(ln.max())} mean {float(ln.float().mean()):.0f}  "
          + " ".join(f"{k}={v:.3f}" for k, v in sorted(stages.items(), key=lambda kv: -kv[1])[:5]), flush=True)
