"""CPU restatement of the reference's training loss (L1 + SSIM) with its analytic gradient.

TEST INFRASTRUCTURE ONLY: imported by tests/ and bench.py's baseline leg, never by the product package.
Pinned against tests/golden/loss/*.npz (recorded from the reference itself) by
tests/test_loss_oracle_golden.py.

Follows reference utils/loss_utils.py:
  gaussian(11, 1.5) :38-53, create_window :56-71 (2-D window = outer product of the 1-D one),
  ssim :74-115 (five zero-padded grouped convolutions, C1 = 0.01^2, C2 = 0.03^2, mean of the map),
  l1_loss :10-21, and train.py:141-146 for how they are combined:
      loss = w_l1 * L1 + w_ssim * SSIM_mean      (the reference uses w_l1 = 1 - lambda, w_ssim = +lambda)
The gradient is the derivative of those lines (autograd in the reference).

Arithmetic is float64 on float32 inputs (outputs cast back): sigma = E[x^2] - mu^2 cancels badly in
float32 and the reference's own float32 result moves by ~3e-7 (SSIM) / 2e-4 of max|grad| against its
float64 run, so the oracle states the value both float32 implementations (reference, HIP) approximate.
"""
from __future__ import annotations

import numpy as np

WINDOW = 11
SIGMA = 1.5
C1 = 0.01 ** 2
C2 = 0.03 ** 2


# The reference's window, float32, exactly as its gaussian(11, 1.5) returns it (utils/loss_utils.py:38-53;
# recorded in tests/golden/loss/_window_1d.npz).  The values matter to the last bit: they sum to
# 1 - 3.1e-8, and because sigma = E[x^2] - mu^2 cancels, a window that sums to 1 + 1e-7 instead moves
# the SSIM of a smooth image by 7e-6.
WINDOW_1D_HEX = ("0x1.0d956cp-10", "0x1.f1fe02p-8", "0x1.26eb18p-5", "0x1.bff0fep-4", "0x1.b43c3ep-3", "0x1.10656p-2",
                 "0x1.b43c3ep-3", "0x1.bff0fep-4", "0x1.26eb18p-5", "0x1.f1fe02p-8", "0x1.0d956cp-10")


def window_1d() -> np.ndarray:
    return np.array([float.fromhex(h) for h in WINDOW_1D_HEX], dtype=np.float32)


def _blur(a: np.ndarray) -> np.ndarray:
    """Zero-padded 11x11 correlation of every channel of a (C,H,W) array with the 2-D window (float64)."""
    g = window_1d()
    w2 = np.outer(g, g).astype(np.float32).astype(np.float64)   # create_window :56-71: float32 outer product
    C, H, W = a.shape
    r = WINDOW // 2
    p = np.zeros((C, H + 2 * r, W + 2 * r), np.float64)
    p[:, r:r + H, r:r + W] = a
    out = np.zeros((C, H, W), np.float64)
    for dy in range(WINDOW):
        for dx in range(WINDOW):
            out += w2[dy, dx] * p[:, dy:dy + H, dx:dx + W]
    return out


def l1_ssim(img: np.ndarray, gt: np.ndarray, w_l1: float, w_ssim: float, with_grad: bool = True):
    """Returns dict(loss, l1, ssim[, grad]) for (C,H,W) float32 images."""
    x = np.ascontiguousarray(img, np.float32).astype(np.float64)
    y = np.ascontiguousarray(gt, np.float32).astype(np.float64)
    M = float(x.size)
    mu1, mu2 = _blur(x), _blur(y)
    e11, e22, e12 = _blur(x * x), _blur(y * y), _blur(x * y)
    s11, s22, s12 = e11 - mu1 * mu1, e22 - mu2 * mu2, e12 - mu1 * mu2
    A1 = 2 * mu1 * mu2 + C1
    A2 = 2 * s12 + C2
    B1 = mu1 * mu1 + mu2 * mu2 + C1
    B2 = s11 + s22 + C2
    S = (A1 * A2) / (B1 * B2)
    l1 = np.abs(x - y).sum(dtype=np.float64) / float(M)
    ssim = S.sum(dtype=np.float64) / float(M)
    out = dict(l1=np.float32(l1), ssim=np.float32(ssim), loss=np.float32(w_l1 * l1 + w_ssim * ssim))
    if with_grad:
        inv = 1.0 / (B1 * B2)
        d_mu1 = 2 * mu2 * (A2 - A1) * inv - S * 2 * mu1 * (B2 - B1) * inv
        d_e11 = -S / B2
        d_e12 = 2 * A1 * inv
        # the window is symmetric, so the adjoint of the zero-padded correlation is the same blur
        gs = _blur(d_mu1) + 2 * x * _blur(d_e11) + y * _blur(d_e12)
        out["grad"] = (w_ssim / M * gs + w_l1 / M * np.sign(x - y)).astype(np.float32)
    return out
