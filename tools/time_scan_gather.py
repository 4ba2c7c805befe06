"""What bounds gs_scan_tiles' box gather at N = 3 M: (a) random order, every rank fetched; (b) the last 15 % / 50 % / all ranks
marked not drawn (their boxes are not fetched); (c) identity order (the gather is coalesced).  usage: python tools/time_scan_gather.py [N]"""
import ctypes as C, importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
L = importlib.import_module("gaussian-splatting_amd._lib").load()
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3_000_000
rng = np.random.default_rng(0)
x0, y0 = rng.integers(0, 1200, n), rng.integers(0, 800, n)
w, h = rng.integers(1, 40, n), rng.integers(1, 40, n)
rects = np.stack([x0 | (x0 + w) << 16, y0 | (y0 + h) << 16], axis=1).astype(np.uint32)
t = lambda a: torch.from_numpy(a.view(np.int32).copy()).to(dev)  # noqa: E731
p = lambda x: None if x is None else C.c_void_p(x.data_ptr())  # noqa: E731
d_rects = t(rects)
d_rio, d_off = torch.empty_like(d_rects), torch.empty((n,), dtype=torch.int32, device=dev)
total = torch.zeros((1,), dtype=torch.int32, device=dev)
ws = torch.empty(max(16, L.gs_scan_workspace_bytes(n)), dtype=torch.uint8, device=dev)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
def run(order, keys, label):
    for _ in range(5):
        L.gs_scan_tiles(n, p(d_rects), p(order), p(keys), p(d_rio), p(d_off), p(total), p(ws), None, 0, 1 << 20, st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        L.gs_scan_tiles(n, p(d_rects), p(order), p(keys), p(d_rio), p(d_off), p(total), p(ws), None, 0, 1 << 20, st)
    e1.record(); torch.cuda.synchronize()
    print(f"{label}: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us")
perm = t(rng.permutation(n).astype(np.uint32))
ident = t(np.arange(n, dtype=np.uint32))
run(perm, None, "random order, every rank fetched")
for frac in (0.15, 0.5, 1.0):
    k = np.zeros(n, dtype=np.uint32); k[n - int(frac * n):] = 0xFFFFFFFF
    run(perm, t(k), f"random order, last {int(frac * 100)} % of the ranks not fetched")
run(ident, None, "identity order (coalesced gather)")
# the same number of random fetches out of a smaller footprint (what a table of 4-byte boxes would be: half the bytes)
for div in (2, 4, 8):
    run(t(rng.integers(0, n // div, n).astype(np.uint32)), None, f"random ranks into the first 1/{div} of the table ({8 * n // div >> 20} MB)")
