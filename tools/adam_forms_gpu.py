"""Which rounding forms do torch's GPU kernels (ROCm build) use for the ops of Adam?  fma emulated in float64."""
import torch
dev = torch.device("cuda:0")
f32, f64 = torch.float32, torch.float64
def fma(a, b, c):
    return (a.to(f64) * b.to(f64) + c.to(f64)).to(f32)
def sc(x):
    return torch.tensor(x, dtype=f32, device=dev)
g = torch.Generator(device="cpu").manual_seed(0)
N = 400000
m = (torch.randn(N, generator=g) * 1e-3).to(dev); gr = (torch.randn(N, generator=g) * 1e-3).to(dev)
v = (torch.rand(N, generator=g) * 1e-6).to(dev); p = torch.randn(N, generator=g).to(dev)
w = 1 - 0.9
for name, fn in (("single", lambda: m.clone().lerp_(gr, w)), ("foreach", lambda: torch._foreach_lerp_([m.clone()], [gr], w))):
    r = fn(); r = r if torch.is_tensor(r) else None
    if r is None:
        t = [m.clone()]; torch._foreach_lerp_(t, [gr], w); r = t[0]
    print("lerp", name, "plain", int((m + (gr - m) * sc(w) != r).sum()), "fma", int((fma(sc(w), gr - m, m) != r).sum()))
a = 1 - 0.999
V = v * sc(0.999)
for name in ("single", "foreach"):
    if name == "single":
        r = v.clone().mul_(0.999).addcmul_(gr, gr, value=a)
    else:
        t = [v.clone()]; torch._foreach_mul_(t, 0.999); torch._foreach_addcmul_(t, [gr], [gr], a); r = t[0]
    print("mul", name, int((V != (v.clone().mul_(0.999))).sum()))
    print("addcmul", name, "(a*g)*g+V", int((V + (sc(a) * gr) * gr != r).sum()), "a*(g*g)+V", int((V + sc(a) * (gr * gr) != r).sum()),
          "fma(a*g,g,V)", int((fma(sc(a) * gr, gr, V) != r).sum()), "fma(a,g*g,V)", int((fma(sc(a), gr * gr, V) != r).sum()))
import math
bc = math.sqrt(1 - 0.999 ** 3)
s_exact = v.to(f64).sqrt().to(f32)
print("sqrt exact", int((v.sqrt() != s_exact).sum()))
t = [v.clone()]; t = torch._foreach_sqrt(t); print("foreach sqrt exact", int((t[0] != s_exact).sum()))
s = v.sqrt()
for name in ("single", "foreach"):
    if name == "single":
        r = s / bc
    else:
        t = [s.clone()]; torch._foreach_div_(t, bc); r = t[0]
    print("div", name, "s/f32(bc)", int((s / sc(bc) != r).sum()), "s*f32(1/bc)", int((s * sc(1.0 / bc) != r).sum()),
          "s*(1/f32(bc))", int((s * (1 / sc(bc)) != r).sum()), "double div", int(((s.to(f64) / bc).to(f32) != r).sum()))
d = (s / bc).add_(1e-15)
val = -0.0025
for name in ("single", "foreach"):
    if name == "single":
        r = p.clone().addcdiv_(m, d, value=val)
    else:
        t = [p.clone()]; torch._foreach_addcdiv_(t, [m], [d], val); r = t[0]
    print("addcdiv", name, "p+(a*m)/d", int((p + (sc(val) * m) / d != r).sum()), "p+a*(m/d)", int((p + sc(val) * (m / d) != r).sum()),
          "fma(a,m/d,p)", int((fma(sc(val), m / d, p) != r).sum()))
