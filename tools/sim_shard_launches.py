"""What the per-view launches of a SplatExchange rank cost at world = W: W x gs_preprocess_fwd and
W x gs_preprocess_bwd (accumulating) over a shard of N / W Gaussians, against one launch over N.
GPU time from events, wall time around the queueing + sync (shows whether the host keeps up)."""
import importlib
import sys
import time

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
R = importlib.import_module("gaussian-splatting_amd.rasterizer")
syn = importlib.import_module("gaussian-splatting_amd.synthetic")
impl = importlib.import_module("gaussian-splatting_amd")

dev = torch.device("cuda:0")
N, K, H, W = 3_000_000, 16, 840, 1297
sc = syn.make_scene(N, K, H, W, seed=0)


def settings(v):
    return impl.GaussianRasterizationSettings(
        image_height=H, image_width=W, tanfovx=v.tanfovx, tanfovy=v.tanfovy, bg=v.bg.to(dev), scale_modifier=1.0,
        viewmatrix=v.viewmatrix.to(dev), projmatrix=v.projmatrix.to(dev), sh_degree=v.sh_degree,
        campos=v.campos.to(dev), prefiltered=False, debug=False)


for world in (1, 2, 4, 8):
    n = N // world
    views = [settings(syn.jitter_view(sc, v)) for v in range(world)]
    m3, sh, op, s, q = (getattr(sc, k)[:n].to(dev) for k in ("means3D", "shs", "opacities", "scales", "rotations"))
    rows = torch.cat([torch.randn(world, n, 6, dtype=torch.float64).view(torch.float32), torch.randn(world, n, 4)], dim=2).contiguous().to(dev)
    needs = [True] * 8

    def step():
        radii = []
        for v in range(world):
            _, r, *_ = R.preprocess(views[v], m3, op, s, q, None, sh, None)
            radii.append(r)
        acc = None
        for v in range(world):
            acc = R.preprocess_backward(views[v], m3, op, s, q, None, sh, None, radii[v], rows[v], needs, (n, 3),
                                        accumulate_into=acc)
        return acc

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    a.record()
    for _ in range(20):
        step()
    b.record()
    tq = time.perf_counter() - t0
    torch.cuda.synchronize()
    tw = time.perf_counter() - t0
    print(f"world {world}: shard {n}  gpu {a.elapsed_time(b) / 20:.3f} ms/step  host queueing {tq / 20 * 1e3:.3f} ms/step  "
          f"wall {tw / 20 * 1e3:.3f} ms/step", flush=True)
