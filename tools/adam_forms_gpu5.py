import importlib, sys, math
import numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from test_adam_oracle import make_grads
impl = importlib.import_module("gaussian-splatting_amd")
dev = torch.device("cuda:0")
f32, f64 = np.float32, np.float64
G = make_grads(777, 101)["opacity"].numpy().ravel()
G = G[G != 0]
lr = 1 - 0.9 ** 1          # step size = lr / bc1 = 1  ->  p = -q when p0 = 0
b = torch.nn.Parameter(torch.zeros(G.size, device=dev)); b.grad = torch.from_numpy(G).to(dev)
impl.FusedAdam([{"params": [b], "lr": lr}], lr=0.0, eps=1e-15).step()
q_fused = -b.detach().cpu().numpy()
def fma(a, b_, c): return (np.asarray(a, f64) * np.asarray(b_, f64) + np.asarray(c, f64)).astype(f32)
M = fma(f32(1 - 0.9), G, np.zeros_like(G)); V = fma(f32(1 - 0.999), G * G, np.zeros_like(G))
bc2s = (1 - 0.999 ** 1) ** 0.5
S = np.sqrt(V)
print("bc2s pow", repr(bc2s), "sqrt", repr(math.sqrt(1 - 0.999 ** 1)), "f32 equal", f32(bc2s) == f32(math.sqrt(1 - 0.999)))
for k, D in {"div": S / f32(bc2s) + f32(1e-15), "recip": S * f32(1 / bc2s) + f32(1e-15)}.items():
    print(k, "q mismatches", (M / D != q_fused).sum(), "of", G.size)
i = np.flatnonzero(M / (S / f32(bc2s) + f32(1e-15)) != q_fused)[:3]
for j in i:
    print("g %r m %r v %r s %r d %r q_exp %r q_fused %r" % (G[j], M[j], V[j], S[j], (S / f32(bc2s) + f32(1e-15))[j], (M / (S / f32(bc2s) + f32(1e-15)))[j], q_fused[j]))
