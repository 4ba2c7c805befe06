"""CPU: what the default stop threshold (t_eps, the HIP path's one switchable deviation from the reference) can cost, as a
stated bound that these tests enforce on the adversarial scenes of tools/fuzz_parity.py.

The oracle (oracle/gs_oracle.c) applies the same stop rule when asked to (gso_set_t_eps): a pixel whose transmittance
1 - alpha_acc has dropped to <= t_eps composites nothing further.  Against the reference's never-stopping compositing
(t_eps = 0, reference gaussian_renderer/__init__.py:429-440), per pixel p and Gaussian i, with
    u_i = c_i . g_C + z_i g_Z + g_A,   U_p = sum_c |g_C| + max|z| |g_Z| + |g_A|   (colours lie in [0, 1]),
  * outputs: the dropped alphas add up to at most the transmittance at the stop, so
        |d image| <= t_eps,   |d alpha| <= t_eps,   |d depth| <= t_eps max|z|;
  * dL/dw_pi of a Gaussian BEHIND the stop: T_i |u_i| <= t_eps U and the part behind it over (1 - w_i) is <= T_i U too:
        |d (dL/dw_pi)| <= 2 t_eps U_p;
  * dL/dw_pi of a Gaussian AT or BEFORE the stop loses (dropped tail) / (1 - w_pi), dropped tail <= t_eps U, w_pi <= o_i,
    and never more than T_i U <= U:
        |d (dL/dw_pi)| <= min(1, t_eps / (1 - o_i)) U_p.
Summed over the pixels of the Gaussian's box (exp(power) <= 1, alpha <= T):
    |d (dL/d opacity_i)| <= max(2 t_eps, min(1, t_eps / (1 - o_i))) * sum_{p in box_i} U_p
    |d (dL/d rgb_i)|     <= t_eps * sum_{p in box_i} |g_C(p)|            (per channel)
The bounds are ABSOLUTE: a nearly opaque Gaussian (o -> 1) in front of a bright tail has a large true derivative through
1 / (1 - w) that the truncation removes -- relative to a small gradient tensor that can be percent-level (DESIGN.md
section 1), which is why t_eps = 0 (GSPLAT_T_EPS=0) is the parity mode.  tests/test_gpu_fuzz.py holds the HIP path at the
default t_eps to the oracle at the SAME t_eps with the usual relative bar; this file states what that mode is worth
against the reference.
"""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))

T_EPS = 1e-5
SLACK = 1.02          # fp32 rounding of the oracle's own sums
SEEDS = list(range(200))


def _box_sums(img, rect):
    """sum of img over every Gaussian's half-open box (x0, y0, x1, y1), via a summed-area table."""
    H, W = img.shape
    sat = np.zeros((H + 1, W + 1), np.float64)
    sat[1:, 1:] = np.cumsum(np.cumsum(img.astype(np.float64), axis=0), axis=1)
    x0, y0, x1, y1 = (rect[:, k].astype(np.int64) for k in range(4))
    return sat[y1, x1] - sat[y0, x1] - sat[y1, x0] + sat[y0, x0]


@pytest.mark.parametrize("seed", SEEDS)
def test_truncation_is_within_the_stated_bound(oracle, seed):
    import fuzz_parity as F
    case = F.make_case(seed)
    outs0, g0, geom = F._oracle_call(oracle, case, 0.0)
    outs1, g1, geom1 = F._oracle_call(oracle, case, T_EPS)
    assert np.array_equal(geom.rect, geom1.rect)
    drawn = geom.tiles_touched > 0
    zmax = float(np.abs(geom.depth[drawn]).max()) if drawn.any() else 0.0
    assert np.abs(outs1["render"] - outs0["render"]).max() <= SLACK * T_EPS + 1e-7
    assert np.abs(outs1["alpha"] - outs0["alpha"]).max() <= SLACK * T_EPS + 1e-7
    assert np.abs(outs1["depth"] - outs0["depth"]).max() <= SLACK * T_EPS * zmax + 1e-6
    gC, gZ, gA = np.abs(case["G_C"]), np.abs(case["G_Z"][0]), np.abs(case["G_A"][0])
    U = gC.sum(axis=0) + zmax * gZ + gA
    rect = np.where(drawn[:, None], geom.rect, 0)
    sumU = _box_sums(U, rect)
    o = case["opacities"].reshape(-1).astype(np.float64)
    factor = np.maximum(2 * T_EPS, np.minimum(1.0, T_EPS / np.maximum(1.0 - o, 1e-300)))
    rb0, rb1 = g0["render_bwd"], g1["render_bwd"]
    d_op = np.abs(rb1["d_opacity"].reshape(-1).astype(np.float64) - rb0["d_opacity"].reshape(-1))
    bound = SLACK * factor * sumU + 1e-6 * np.abs(rb0["d_opacity"].reshape(-1))
    bad = np.nonzero(d_op > bound)[0]
    assert bad.size == 0, f"dL/d opacity of Gaussian {bad[0]}: off by {d_op[bad[0]]:.3e} > bound {bound[bad[0]]:.3e}"
    for c in range(3):
        d_c = np.abs(rb1["d_rgb"][:, c].astype(np.float64) - rb0["d_rgb"][:, c])
        bc = SLACK * T_EPS * _box_sums(gC[c], rect) + 1e-6 * np.abs(rb0["d_rgb"][:, c])
        bad = np.nonzero(d_c > bc)[0]
        assert bad.size == 0, f"dL/d rgb[{c}] of Gaussian {bad[0]}: off by {d_c[bad[0]]:.3e} > bound {bc[bad[0]]:.3e}"


def test_oracle_threshold_zero_is_the_reference(oracle):
    """t_eps = 0 takes the oracle's unmodified path (bit-identical outputs and gradients)."""
    import fuzz_parity as F
    case = F.make_case(7)
    a = F._oracle_call(oracle, case, 0.0)
    oracle._set_t_eps(0.0)
    assert oracle.lib().gso_get_t_eps() == 0.0
    b = F._oracle_call(oracle, case, 0.0)
    for k in ("render", "depth", "alpha"):
        assert np.array_equal(a[0][k], b[0][k])
    for k in F.INPUTS:
        if a[1].get(k) is not None:
            assert np.array_equal(a[1][k], b[1][k])
