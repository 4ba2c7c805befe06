import importlib, sys, math
import numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from test_adam_oracle import GROUP_LRS, make_grads, make_params
from oracle import adam_oracle
impl = importlib.import_module("gaussian-splatting_amd")
dev = torch.device("cuda:0")
params = make_params(777, 0); grads = make_grads(777, 101)
name, lr = "opacity", 0.05
a = torch.nn.Parameter(params[name].clone().to(dev)); b = torch.nn.Parameter(params[name].clone().to(dev))
a.grad = grads[name].to(dev); b.grad = grads[name].to(dev)
oa = torch.optim.Adam([{"params": [a], "lr": lr}], lr=0.0, eps=1e-15); oa.step()
ob = impl.FusedAdam([{"params": [b], "lr": lr}], lr=0.0, eps=1e-15); ob.step()
op = params[name].numpy().copy(); om = np.zeros_like(op); ov = np.zeros_like(op)
adam_oracle.adam_step(op, grads[name].numpy(), om, ov, 1, lr)
A, B = a.detach().cpu().numpy(), b.detach().cpu().numpy()
print("torch vs fused", (A != B).sum(), "torch vs oracle", (A != op).sum(), "fused vs oracle", (B != op).sum())
i = np.flatnonzero((A != B).ravel())[0]
p0, g = params[name].numpy().ravel()[i], grads[name].numpy().ravel()[i]
print("p0 %r g %r torch %r fused %r oracle %r" % (p0, g, A.ravel()[i], B.ravel()[i], op.ravel()[i]))
print("m torch %r fused %r  v torch %r fused %r" % (oa.state[a]["exp_avg"].flatten()[i].item(), ob.state[b]["exp_avg"].flatten()[i].item(),
      oa.state[a]["exp_avg_sq"].flatten()[i].item(), ob.state[b]["exp_avg_sq"].flatten()[i].item()))
f32 = np.float32
m, v = om.ravel()[i], ov.ravel()[i]
bc1 = 1 - 0.9 ** 1; bc2s = (1 - 0.999 ** 1) ** 0.5
print("neg step double %r f32 %r" % ((lr / bc1) * -1, f32((lr / bc1) * -1)))
s = np.sqrt(v); d = s / f32(bc2s) + f32(1e-15)
print("s %r d %r q %r" % (s, d, m / d))
# which emulated form reproduces the FUSED kernel?
P0 = params[name].numpy().ravel(); G = grads[name].numpy().ravel(); Bf = B.ravel()
f64 = np.float64
def fma(a, b, c): return (np.asarray(a, f64) * np.asarray(b, f64) + np.asarray(c, f64)).astype(f32)
M = fma(f32(1 - 0.9), G - 0, np.zeros_like(G)); V = fma(f32(1 - 0.999), G * G, np.zeros_like(G))
S = np.sqrt(V); ns = f32((lr / bc1) * -1)
forms = {
 "div": S / f32(bc2s) + f32(1e-15),
 "mul recip(double)": S * f32(1.0 / bc2s) + f32(1e-15),
 "mul recip(f32)": S * (f32(1) / f32(bc2s)) + f32(1e-15),
 "div no eps": S / f32(bc2s),
}
for k, D in forms.items():
    with np.errstate(all="ignore"):
        Q = M / D
        print(k, "fma:", (fma(ns, Q, P0) != Bf).sum(), " plain:", ((P0 + ns * Q) != Bf).sum(), " (ns*M)/D:", ((P0 + (ns * M) / D) != Bf).sum())
