"""CPU: the nearest-neighbour oracle (oracle/knn_oracle.py) against the reference's own sequence of torch calls
(scene/gaussian_model.py:734-746, restated in the oracle module: torch.cdist is the dependency the algorithm lives
in) and against scipy's k-d tree."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import knn_oracle  # noqa: E402

# cdist's matmul form loses about eps * |x|^2 in the squared distance; the comparison is made in those units
TOL_CDIST = 4e-6


def cloud(n, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    c = torch.randn(8, 3, generator=g) * 3
    return (c[torch.randint(0, 8, (n,), generator=g)] + torch.randn(n, 3, generator=g) * 0.4) * scale


def test_oracle_vs_reference_form_and_kdtree():
    for n, seed in ((2, 0), (20, 1), (600, 2), (3000, 3)):     # <= 25 points: cdist computes differences directly
        p = cloud(n, seed)
        want = knn_oracle.reference_form(p).numpy().astype(np.float64)
        got = knn_oracle.nearest_dist2(p.numpy())
        lim = TOL_CDIST * float((p ** 2).sum(1).max()) if n > 25 else 1e-6 * want.max()
        assert np.abs(got - want).max() <= lim, (n, np.abs(got - want).max(), lim)
        assert np.allclose(got, knn_oracle.nearest_dist2_kdtree(p.numpy()), rtol=1e-12, atol=0)


def test_degenerate_inputs():
    assert knn_oracle.nearest_dist2(np.zeros((1, 3)))[0] == np.inf
    assert float(knn_oracle.reference_form(torch.zeros(1, 3))[0]) == float("inf")
    d = knn_oracle.nearest_dist2(np.array([[0, 0, 0], [0, 0, 0], [1, 0, 0]], float))
    assert d.tolist() == [0.0, 0.0, 1.0]
