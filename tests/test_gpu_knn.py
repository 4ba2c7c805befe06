"""GPU: distCUDA2 (gs_nearest_dist2 through the C-ABI) against the float64 oracle.

Tolerance: the kernel finds the exact nearest neighbour and computes its squared distance from float32 coordinate
differences, so |d2 - oracle| <= 4 eps32 * (d2 + |x| * sqrt(d2)) (rounding of the differences); checked as
rel 1e-5 of d2 + 1e-6 * max|x| * sqrt(d2).  The reference's own value carries cdist's larger cancellation error
(tests/test_knn_oracle.py)."""
import importlib
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import knn_oracle  # noqa: E402
from test_knn_oracle import cloud  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def knn():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return importlib.import_module("gaussian-splatting_amd.knn")


def _check(got, pts, want):
    got = got.cpu().numpy().astype(np.float64)
    fin = np.isfinite(want)
    assert np.array_equal(np.isfinite(got), fin)
    xmax = float(np.abs(pts).max()) if pts.size else 0.0
    lim = 1e-5 * want[fin] + 1e-6 * xmax * np.sqrt(want[fin]) + 1e-30
    err = np.abs(got[fin] - want[fin])
    assert np.all(err <= lim), (err.max(), int((err > lim).sum()))


@pytest.mark.parametrize("n,seed", [(1, 0), (2, 1), (3, 2), (100, 3), (5000, 4), (20000, 5)])
def test_clustered_cloud_vs_oracle(knn, n, seed):
    p = cloud(n, seed)
    d, st = knn.distCUDA2(p.cuda(), return_stats=True)
    assert d.shape == (n,)
    _check(d, p.numpy(), knn_oracle.nearest_dist2(p.numpy()))


def test_outliers_duplicates_planar_and_offsets(knn):
    g = torch.Generator().manual_seed(7)
    p = cloud(8000, 9)
    p[:5] = torch.tensor([[500.0, 0, 0], [0, -700.0, 0], [0, 0, 900.0], [400.0, 400.0, 400.0], [-300.0, 2.0, 5.0]])  # isolated
    p[100:110] = p[100]                                             # duplicates -> 0
    d, st = knn.distCUDA2(p.cuda(), return_stats=True)
    assert st["isolated"] >= 5                                      # the brute-force pass really ran
    _check(d, p.numpy(), knn_oracle.nearest_dist2(p.numpy()))
    assert float(d[100:110].max()) == 0.0
    flat = torch.rand(6000, 3, generator=g) * torch.tensor([10.0, 10.0, 0.0]) + torch.tensor([1e3, -2e3, 5.0])  # planar, far from 0
    _check(knn.distCUDA2(flat.cuda()), flat.numpy(), knn_oracle.nearest_dist2(flat.numpy()))
    same = torch.ones(300, 3) * 2.5                                 # zero extent
    assert float(knn.distCUDA2(same.cuda()).max()) == 0.0
    line = torch.zeros(4000, 3)
    line[:, 0] = torch.rand(4000, generator=g) * 100
    _check(knn.distCUDA2(line.cuda()), line.numpy(), knn_oracle.nearest_dist2(line.numpy()))


@pytest.mark.parametrize("n,frac,sigma", [(200000, 0.9, 0.01), (100000, 0.99, 0.001), (150000, 0.5, 0.1)])
def test_very_uneven_density_takes_the_adaptive_pass(knn, n, frac, sigma):
    """SfM-like clouds: most points in a few tight clusters inside a large box.  The uniform grid alone is quadratic
    there; crowded points go through the Morton-ordered adaptive pass (stats["crowded"]) and stay exact."""
    g = torch.Generator().manual_seed(int(1000 * sigma) + n)
    nd = int(n * frac)
    centres = (torch.rand(10, 3, generator=g) - 0.5) * 20
    dense = centres[torch.randint(0, 10, (nd,), generator=g)] + torch.randn(nd, 3, generator=g) * sigma
    p = torch.cat([dense, (torch.rand(n - nd, 3, generator=g) - 0.5) * 20])
    p[:50] = p[50:100]                                              # duplicates inside the crowded part
    d, st = knn.distCUDA2(p.cuda(), return_stats=True)
    assert st["crowded"] > 0.3 * nd
    _check(d, p.numpy(), knn_oracle.nearest_dist2_kdtree(p.numpy()))
    assert float(d[:100].max()) == 0.0


def test_large_uniform_vs_kdtree_and_errors(knn):
    g = torch.Generator().manual_seed(11)
    p = torch.rand(400000, 3, generator=g) * torch.tensor([20.0, 12.0, 6.0])
    d, st = knn.distCUDA2(p.cuda(), return_stats=True)
    _check(d, p.numpy(), knn_oracle.nearest_dist2_kdtree(p.numpy()))
    assert st["isolated"] < 100 and st["crowded"] == 0
    assert knn.distCUDA2(torch.zeros(0, 3).cuda()).shape == (0,)
    with pytest.raises(RuntimeError):
        knn.distCUDA2(torch.zeros(4, 3))
    with pytest.raises(ValueError):
        knn.distCUDA2(torch.zeros(4, 2).cuda())
