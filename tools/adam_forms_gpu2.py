import importlib, sys, math
import torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from test_adam_oracle import GROUP_LRS, make_grads, make_params
impl = importlib.import_module("gaussian-splatting_amd")
dev = torch.device("cuda:0")
f32, f64 = torch.float32, torch.float64
g = torch.Generator().manual_seed(0)
N = 400000
m = (torch.randn(N, generator=g) * 1e-3).to(dev); d = (torch.rand(N, generator=g) * 1e-2 + 1e-5).to(dev)
q = m / d
exact = (m.to(f64) / d.to(f64)).to(f32)
print("torch div IEEE mismatches", int((q != exact).sum()))
# scalar-list foreach variants
bc = math.sqrt(1 - 0.999 ** 1)
s = (torch.rand(N, generator=g) * 1e-3).to(dev)
t = [s.clone()]; torch._foreach_div_(t, [bc]); r = t[0]
sc = lambda x: torch.tensor(x, dtype=f32, device=dev)
print("foreach_div scalarlist: s/f32(bc)", int((s / sc(bc) != r).sum()), "s*f32(1/bc)", int((s * sc(1.0 / bc) != r).sum()),
      "IEEE s/f32(bc)", int(((s.to(f64) / sc(bc).to(f64)).to(f32) != r).sum()))
# the failing case
params = make_params(777, 0); grads = make_grads(777, 101)
for name, _, lr in GROUP_LRS:
    a = torch.nn.Parameter(params[name].clone().to(dev)); b = torch.nn.Parameter(params[name].clone().to(dev))
    a.grad = grads[name].to(dev); b.grad = grads[name].to(dev)
    torch.optim.Adam([{"params": [a], "lr": lr}], lr=0.0, eps=1e-15).step()
    impl.FusedAdam([{"params": [b], "lr": lr}], lr=0.0, eps=1e-15).step()
    c = torch.nn.Parameter(params[name].clone().to(dev)); c.grad = grads[name].to(dev)
    torch.optim.Adam([{"params": [c], "lr": lr}], lr=0.0, eps=1e-15, foreach=False).step()
    bad = (a != b).flatten().nonzero().flatten()
    print(name, "fused vs foreach:", int(bad.numel()), " fused vs single-tensor:", int((c != b).sum()), " foreach vs single:", int((a != c).sum()))
    if bad.numel():
        i = int(bad[0]); print("   p0", params[name].flatten()[i].item(), "g", grads[name].flatten()[i].item(), "torch", a.flatten()[i].item(), "fused", b.flatten()[i].item())
