"""distCUDA2 on strongly non-uniform clouds (SfM-like): time and grid statistics."""
import importlib, sys, time, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
knn = importlib.import_module("gaussian-splatting_amd.knn")
g = torch.Generator().manual_seed(0)
def cloud(n, frac_dense, sigma):
    nd = int(n * frac_dense)
    centres = (torch.rand(10, 3, generator=g) - 0.5) * 20
    dense = centres[torch.randint(0, 10, (nd,), generator=g)] + torch.randn(nd, 3, generator=g) * sigma
    rest = (torch.rand(n - nd, 3, generator=g) - 0.5) * 20
    return torch.cat([dense, rest])
for n, frac, sigma in ((1_000_000, 0.0, 1.0), (1_000_000, 0.5, 0.1), (1_000_000, 0.9, 0.01), (1_000_000, 0.99, 0.001), (3_000_000, 0.9, 0.01)):
    p = cloud(n, frac, sigma).cuda()
    d, st = knn.distCUDA2(p, return_stats=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    d = knn.distCUDA2(p)
    torch.cuda.synchronize()
    print(f"N {n} dense {frac} sigma {sigma}: {1e3 * (time.perf_counter() - t0):.1f} ms  grid {st['grid']} isolated {st['isolated']}", flush=True)
