"""CPU oracle of the nearest-neighbour distance used for the scale initialisation.
TEST INFRASTRUCTURE: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

Restates distCUDA2 (reference scene/gaussian_model.py:729-746): pairwise Euclidean distances, diagonal = inf,
row minimum, squared.  Computed here from coordinate differences in float64 (chunked brute force, O(N) memory);
`reference_form` is the reference's own sequence of torch calls, used by tests/test_knn_oracle.py to pin the
oracle: torch.cdist switches to |a|^2 + |b|^2 - 2 a.b for more than 25 points, whose float32 cancellation error
(about 1e-6 * max|x|^2 in the squared distance) is the stated tolerance of that comparison.
"""
import numpy as np


def nearest_dist2(points, chunk=2048):
    p = np.asarray(points, dtype=np.float64)
    n = p.shape[0]
    out = np.full(n, np.inf)
    for s in range(0, n, chunk):
        d = ((p[s:s + chunk, None, :] - p[None, :, :]) ** 2).sum(-1)
        d[np.arange(d.shape[0]), np.arange(s, s + d.shape[0])] = np.inf
        out[s:s + chunk] = d.min(1)
    return out


def nearest_dist2_kdtree(points):
    """Independent check for sizes the brute force cannot do."""
    from scipy.spatial import cKDTree
    p = np.asarray(points, dtype=np.float64)
    if p.shape[0] < 2:
        return np.full(p.shape[0], np.inf)
    d, _ = cKDTree(p).query(p, k=2)
    return d[:, 1] ** 2


def reference_form(points_t):
    """The reference's lines (:734-746) on a torch tensor."""
    import torch
    points_t = points_t.contiguous()
    distances = torch.cdist(points_t, points_t, p=2)
    distances.fill_diagonal_(float("inf"))
    return torch.min(distances, dim=1)[0] ** 2
