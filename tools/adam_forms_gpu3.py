"""Rounding forms of the foreach (scalar-list) kernels _multi_tensor_adam uses on the GPU."""
import math
import torch
dev = torch.device("cuda:0")
f32, f64 = torch.float32, torch.float64
def fma(a, b, c):
    return (a.to(f64) * b.to(f64) + c.to(f64)).to(f32)
sc = lambda x: torch.tensor(x, dtype=f32, device=dev)
g = torch.Generator().manual_seed(0)
N = 400000
m = (torch.randn(N, generator=g) * 1e-3).to(dev); gr = (torch.randn(N, generator=g) * 1e-3).to(dev)
v = (torch.rand(N, generator=g) * 1e-6).to(dev); p = torch.randn(N, generator=g).to(dev)
# lerp / mul / addcmul as _multi_tensor_adam calls them
t = [m.clone()]; torch._foreach_lerp_(t, [gr], 1 - 0.9); print("lerp fma", int((fma(sc(1 - 0.9), gr - m, m) != t[0]).sum()))
t = [v.clone()]; torch._foreach_mul_(t, 0.999); V = t[0].clone(); print("mul", int((v * sc(0.999) != V).sum()))
torch._foreach_addcmul_(t, [gr], [gr], 1 - 0.999); r = t[0]
print("addcmul fma(a,g*g,V)", int((fma(sc(1 - 0.999), gr * gr, V) != r).sum()))
s = torch._foreach_sqrt([r])[0]
print("sqrt IEEE", int((r.to(f64).sqrt().to(f32) != s).sum()))
bc = (1 - 0.999 ** 3) ** 0.5
t = [s.clone()]; torch._foreach_div_(t, [bc]); d0 = t[0].clone()
print("div scalarlist IEEE s/f32(bc)", int(((s.to(f64) / sc(bc).to(f64)).to(f32) != d0).sum()), " s/double(bc)", int(((s.to(f64) / bc).to(f32) != d0).sum()))
torch._foreach_add_(t, 1e-15); d = t[0]
print("add eps", int((d0 + sc(1e-15) != d).sum()))
step = (0.0025 / (1 - 0.9 ** 3)) * -1
t = [p.clone()]; torch._foreach_addcdiv_(t, [m], [d], [step]); r = t[0]
q = (m.to(f64) / d.to(f64)).to(f32)
print("addcdiv scalarlist: fma(a,m/d,p)", int((fma(sc(step), q, p) != r).sum()), " p+a*(m/d)", int((p + sc(step) * q != r).sum()),
      " p+(a*m)/d", int((p + ((sc(step) * m).to(f64) / d.to(f64)).to(f32) != r).sum()),
      " double all", int(((p.to(f64) + step * (m.to(f64) / d.to(f64))).to(f32) != r).sum()),
      " fma(double a)", int(((torch.tensor(step, dtype=f64, device=dev) * q.to(f64) + p.to(f64)).to(f32) != r).sum()))
t = [p.clone()]; torch._foreach_addcdiv_(t, [m], [d], torch.tensor([step], dtype=f32)); r2 = t[0]
print("tensor-scalars variant == scalarlist", int((r2 != r).sum()))
