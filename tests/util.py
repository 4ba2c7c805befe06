"""Shared helpers for the parity tests: golden loading and the stated tolerances.

Stated fp32 tolerance (the reference's own fp32-vs-fp64 noise floor is
image 5e-6, depth 4e-5, gradients 5e-6*max|grad|):
  image, alpha : max-abs <= 1e-4
  depth        : max-abs <= 1e-3   (values reach ~10)
  gradients    : per tensor, max-abs <= 1e-3 * max|grad_ref|
plus, against the torch reference only, a budget of bounding-box-edge outlier pixels
(the integer 3-sigma box is a hard cut worth up to exp(-4.5)*opacity ~ 0.011 and a 1-ulp
difference in the covariance can move one edge by one pixel): at most 1e-4 of the pixels
(at least 2) may differ by up to 0.012 * (number of such edges it can sit on, 2).
"""
import glob
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

TOL_IMAGE = 1e-4
TOL_DEPTH = 1e-3
TOL_GRAD_REL = 1e-3
OUTLIER_FRAC = 1e-4
OUTLIER_MAG = 0.024


def golden_names():
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))


def load_golden(name):
    with np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False) as z:
        d = {k: z[k] for k in z.files}
    for k in ("H", "W", "sh_degree", "tanfovx", "tanfovy"):
        d[k] = d[k].reshape(()).item()
    return d


def golden_inputs(g):
    """kwargs of the rasterizer call held by a golden file (numpy arrays, None where absent)."""
    return dict(
        means3D=g["in_means3D"], opacities=g["in_opacities"],
        scales=g.get("in_scales"), rotations=g.get("in_rotations"),
        cov3D_precomp=g.get("in_cov3D_precomp"), shs=g.get("in_shs"),
        colors_precomp=g.get("in_colors_precomp"))


def check_image(name, got, ref, tol, scale=1.0):
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, f"{name}: shape {got.shape} vs {ref.shape}"
    err = np.abs(got - ref)
    bad = err > tol
    nbad = int(bad.sum())
    budget = max(2, int(OUTLIER_FRAC * err.size))
    assert nbad <= budget, f"{name}: {nbad} px over {tol} (budget {budget}), max err {err.max():.3e}"
    if nbad:
        assert err.max() <= OUTLIER_MAG * scale, f"{name}: outlier {err.max():.3e} beyond a bbox-edge flip"
    return float(err.max())


def check_image_strict(name, got, ref, tol):
    """max-abs <= tol at every pixel, no outlier budget: for HIP against the C oracle, whose boxes are bit-identical."""
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, f"{name}: shape {got.shape} vs {ref.shape}"
    assert np.isfinite(got).all(), f"{name}: non-finite"
    err = np.abs(got - ref).max() if got.size else 0.0
    assert err <= tol, f"{name}: max-abs err {err:.3e} > {tol}"
    return float(err)


def check_grad(name, got, ref, rel=TOL_GRAD_REL):
    got = np.asarray(got, dtype=np.float64).reshape(-1)
    ref = np.asarray(ref, dtype=np.float64).reshape(-1)
    assert got.shape == ref.shape, f"{name}: shape {got.shape} vs {ref.shape}"
    m = np.abs(ref).max()
    err = np.abs(got - ref).max()
    assert np.isfinite(got).all(), f"{name}: non-finite gradient"
    assert err <= rel * m + 1e-12, f"{name}: max-abs err {err:.3e} > {rel} * max|ref| ({m:.3e})"
    return err / (m + 1e-30)
