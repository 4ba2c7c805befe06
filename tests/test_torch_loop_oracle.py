"""CPU: the PyTorch-loop restatement timed by bench.py's cpu_baseline leg (oracle/torch_loop_oracle.py) reproduces the
golden vectors recorded from the reference (degree-0 cases), outputs and autograd gradients, at the stated tolerances."""
import numpy as np
import pytest
import torch

from util import TOL_DEPTH, TOL_IMAGE, check_grad, check_image, load_golden


@pytest.mark.parametrize("name", ["s64_deg0_seed0", "s64_deg0_seed1"])
def test_torch_loop_matches_reference_golden(name):
    from oracle import torch_loop_oracle as tl
    g = load_golden(name)
    H, W = int(g["H"]), int(g["W"])
    t = lambda a: torch.tensor(np.asarray(a, dtype=np.float32))  # noqa: E731
    leaves = {k: t(g["in_" + k]).requires_grad_(True) for k in ("means3D", "shs", "opacities", "scales", "rotations")}
    img, dep, alp = tl.render(leaves["means3D"], leaves["shs"][:, 0, :], leaves["opacities"], leaves["scales"],
                              leaves["rotations"], t(g["viewmatrix"]).reshape(4, 4), t(g["projmatrix"]).reshape(4, 4),
                              t(g["bg"]), H, W)
    ((img * t(g["G_C"])).sum() + (dep * t(g["G_Z"])).sum() + (alp * t(g["G_A"])).sum()).backward()
    check_image(name + ":render", img.detach().numpy(), g["out_render"], TOL_IMAGE)
    check_image(name + ":alpha", alp.detach().numpy(), g["out_alpha"], TOL_IMAGE)
    check_image(name + ":depth", dep.detach().numpy(), g["out_depth"], TOL_DEPTH, scale=10.0)
    for k in leaves:
        check_grad(f"{name}:d_{k}", leaves[k].grad.numpy(), g["grad_" + k])
