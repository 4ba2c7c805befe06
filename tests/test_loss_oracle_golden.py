"""The CPU loss oracle (oracle/loss_oracle.py) against the vectors recorded from the reference's own
l1_loss / ssim_loss / autograd (tests/golden/gen_golden_loss.py)."""
import glob
import os

import numpy as np
import pytest

from oracle import loss_oracle

GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "loss", "[a-z]*.npz")))

# the reference's own float32 noise against its float64 run on these cases: 3.5e-7 (SSIM), 2e-4 of max|grad|
TOL_LOSS = 1e-6          # absolute, loss values are O(0.1 .. 1)
TOL_GRAD_REL = 3e-4      # of max|grad|


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p)[:-4] for p in GOLD])
def test_loss_oracle_matches_reference(path):
    g = np.load(path)
    lam = float(g["lam"])
    out = loss_oracle.l1_ssim(g["img"], g["gt"], 1.0 - lam, lam)      # train.py:143-145: + lambda * SSIM (sic)
    assert abs(float(out["l1"]) - float(g["l1"])) <= TOL_LOSS
    assert abs(float(out["ssim"]) - float(g["ssim"])) <= TOL_LOSS
    assert abs(float(out["loss"]) - float(g["loss"])) <= TOL_LOSS
    err = np.abs(out["grad"] - g["grad"]).max()
    assert err <= TOL_GRAD_REL * np.abs(g["grad"]).max(), err


def test_window_is_the_references_bit_for_bit():
    w = loss_oracle.window_1d()
    ref = np.load(os.path.join(os.path.dirname(__file__), "golden", "loss", "_window_1d.npz"))["g"]
    assert w.dtype == np.float32 and np.array_equal(w, ref) and np.array_equal(w, w[::-1])
