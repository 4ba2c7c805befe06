#!/usr/bin/env python3
"""Randomised hunt for distCUDA2: mixtures of uniform boxes, tight clusters (the Morton / octree pass), duplicates,
lines, planes, far outliers and large offsets, against the float64 k-d tree oracle (oracle/knn_oracle.py) with the
bar of tests/test_gpu_knn.py.  Test infrastructure.   usage: python tools/fuzz_knn.py [first_seed] [cases]"""
import importlib
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import knn_oracle  # noqa: E402


def make_cloud(seed):
    r = np.random.RandomState(seed)
    n = int(r.choice([2, 3, 17, 64, 65, 1000, 9000, 40000, 150000]))
    parts, left = [], n
    while left > 0:
        m = left if len(parts) >= 5 else int(min(left, max(1, r.randint(1, n + 1))))
        kind = r.randint(6)
        c = r.randn(3) * 10.0 ** r.randint(-1, 3)
        if kind == 0:
            p = c + (r.rand(m, 3) - 0.5) * 10.0 ** r.randint(-2, 2)
        elif kind == 1:
            p = c + r.randn(m, 3) * 10.0 ** r.randint(-5, -1)            # tight cluster
        elif kind == 2:
            p = np.repeat(c[None], m, 0)                                 # duplicates
        elif kind == 3:
            p = c + np.outer(r.rand(m), r.randn(3))                      # line
        elif kind == 4:
            p = c + r.rand(m, 3) * np.array([1.0, 1.0, 0.0])             # plane
        else:
            p = c + r.randn(m, 3) * np.array([1.0, 1e-3, 10.0])          # anisotropic
        parts.append(p)
        left -= m
    p = np.concatenate(parts).astype(np.float32)
    return p[r.permutation(n)]


def check(got, pts, want):
    fin = np.isfinite(want)
    if not np.array_equal(np.isfinite(got), fin):
        return "finite masks differ"
    xmax = float(np.abs(pts).max())
    lim = 1e-5 * want[fin] + 1e-6 * xmax * np.sqrt(want[fin]) + 1e-30
    err = np.abs(got[fin] - want[fin])
    bad = err > lim
    return None if not bad.any() else f"{int(bad.sum())} of {fin.sum()} beyond the bar, worst {float((err / lim).max()):.1f}x"


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    cases = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    knn = importlib.import_module("gaussian-splatting_amd.knn")
    nbad = 0
    for seed in range(first, first + cases):
        p = make_cloud(seed)
        pc = torch.from_numpy(p).cuda()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        d, st = knn.distCUDA2(pc, return_stats=True)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3
        msg = check(d.cpu().numpy().astype(np.float64), p, knn_oracle.nearest_dist2_kdtree(p))
        nbad += msg is not None
        print(f"{'ok  ' if msg is None else 'FAIL'} seed {seed}: n={p.shape[0]} {ms:.1f} ms {st} {msg or ''}", flush=True)
    print(f"{cases - nbad} of {cases} clouds within the bar", flush=True)
    return 1 if nbad else 0


if __name__ == "__main__":
    sys.exit(main())
