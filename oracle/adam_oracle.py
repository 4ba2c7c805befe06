"""CPU oracle of the optimizer step + densification statistics.  TEST INFRASTRUCTURE:
only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

Restates, in numpy float32 and operation by operation:
  * torch.optim.Adam's single-tensor update (torch/optim/adam.py, torch 2.10; no weight decay, amsgrad or
    maximize) as the reference constructs it -- Adam(l, lr=0.0, eps=1e-15) over six groups,
    reference scene/gaussian_model.py:264-297, stepped at train.py:217;
  * max_radii2D[vis] = max(max_radii2D[vis], radii[vis])        (reference train.py:184-187)
  * xyz_gradient_accum[vis] += |viewspace_grad[vis, :2]|; denom[vis] += 1   (scene/gaussian_model.py:710-713)

Pinned (tests/test_adam_oracle.py) against torch.optim.Adam itself -- the third-party dependency the
algorithm lives in, torch 2.10.0 as installed -- at the reference's call-site arguments (group learning
rates of arguments/__init__.py:90-97, eps 1e-15), bit for bit over several steps (with torch's own CPU sqrt
passed in; with the IEEE sqrt the parameters agree to one ulp, the moments exactly).  The three statistic lines
are checked against a torch restatement written from the cited lines; the reference itself could not be run
for this row ("Denial recorded"), so they are "parity unpinned" at the reference.
"""
import math

import numpy as np

f32 = np.float32


def _fma(a, b, c):
    """float32 fma through float64 (the product of two float32 is exact there)."""
    return (np.asarray(a, np.float64) * np.asarray(b, np.float64) + np.asarray(c, np.float64)).astype(f32)


def adam_step(param, grad, exp_avg, exp_avg_sq, step, lr, beta1=0.9, beta2=0.999, eps=1e-15, form="gpu", sqrt=np.sqrt):
    """In place on float32 arrays; `step` is the 1-based count of this update.

    torch's CPU and GPU kernels round the same formulas differently; both forms were determined op by op
    against torch 2.10 (tools/adam_forms_gpu.py, tools/adam_forms_gpu3.py for the GPU; tests/test_adam_oracle.py pins the CPU form):
                 CPU                                    GPU (what the reference runs when it trains on a GPU)
      lerp_      fma(w, g - m, m)                       fma(w, g - m, m)
      addcmul_   fma(a * g, g, v * beta2)               fma(a, g * g, v * beta2)
      denom      sqrt(v) / f32(sqrt(bc2)) + eps         the same (foreach path, torch's default on a GPU; its
                                                        single-tensor GPU path multiplies by f32(1 / sqrt(bc2)))
      addcdiv_   p + (s * m) / d                        fma(s, m / d, p)
    form="cpu" reproduces torch on the CPU bit for bit; "gpu" is the form the HIP kernel follows.
    `sqrt`: torch's CPU sqrt (MKL VML) is not correctly rounded -- 0.6 % of its float32 results are one
    ulp off the IEEE value that numpy and the GPU return -- so the bit-for-bit pin passes torch's in."""
    assert param.dtype == exp_avg.dtype == exp_avg_sq.dtype == np.float32
    g = np.asarray(grad, dtype=f32)
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    step_size = lr / bc1
    bc2_sqrt = math.sqrt(bc2)
    exp_avg[...] = _fma(f32(1.0 - beta1), g - exp_avg, exp_avg)                       # lerp_, weight < 0.5
    exp_avg_sq *= f32(beta2)
    if form == "cpu":
        exp_avg_sq[...] = _fma(f32(1.0 - beta2) * g, g, exp_avg_sq)                   # addcmul_
        denom = sqrt(exp_avg_sq) / f32(bc2_sqrt) + f32(eps)
        param += (f32(-step_size) * exp_avg) / denom                                  # addcdiv_
    else:
        exp_avg_sq[...] = _fma(f32(1.0 - beta2), g * g, exp_avg_sq)
        denom = sqrt(exp_avg_sq) / f32(bc2_sqrt) + f32(eps)
        param[...] = _fma(f32(-step_size), exp_avg / denom, param)
    return param


def densify_stats(means2D_grad, radii, xyz_gradient_accum, denom, max_radii2D):
    """means2D_grad (views,N,3) or None, radii (views,N) int; in place on float32 (N,) arrays, views in order."""
    radii = np.asarray(radii)
    for v in range(radii.shape[0]):
        vis = radii[v] > 0
        if max_radii2D is not None:
            max_radii2D[vis] = np.maximum(max_radii2D[vis], radii[v][vis].astype(f32))
        if means2D_grad is not None:
            g = np.asarray(means2D_grad[v], dtype=f32)
            n = np.sqrt(g[:, 0] * g[:, 0] + g[:, 1] * g[:, 1])
            xyz_gradient_accum[vis] += n[vis]
            denom[vis] += f32(1.0)
