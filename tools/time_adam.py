"""Times gs_adam_densify_step on 3 M Gaussians' worth of parameters (library chosen by GSPLAT_LIB)."""
import importlib, os, sys, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
impl = importlib.import_module("gaussian-splatting_amd")
dev = torch.device("cuda:0")
N, K = 3_000_000, 16
g_shs = torch.randn(N, K, 3, device=dev) * 1e-3
groups = []
for name, shape, lr, grad in (("xyz", (N, 3), 1.6e-4, None), ("f_dc", (N, 1, 3), 2.5e-3, g_shs[:, :1]), ("f_rest", (N, K - 1, 3), 1.25e-4, g_shs[:, 1:]),
                              ("opacity", (N, 1), 0.05, None), ("scaling", (N, 3), 5e-3, None), ("rotation", (N, 4), 1e-3, None)):
    p = torch.nn.Parameter(torch.randn(*shape, device=dev))
    p.grad = grad if grad is not None else torch.randn(*shape, device=dev) * 1e-3
    groups.append({"params": [p], "lr": lr, "name": name})
opt = impl.FusedAdam(groups, lr=0.0, eps=1e-15)
for _ in range(3):
    opt.step()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20):
    opt.step()
b.record(); torch.cuda.synchronize()
ms = a.elapsed_time(b) / 20
print(os.environ.get("GSPLAT_LIB", "default"), f"{ms:.4f} ms  {28 * 177e6 / (ms * 1e-3) / 1e9:.0f} GB/s", flush=True)
