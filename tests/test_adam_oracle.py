"""CPU: the optimizer oracle (oracle/adam_oracle.py) against torch.optim.Adam itself -- the dependency the
reference's optimizer step lives in -- constructed as the reference constructs it
(scene/gaussian_model.py:264-297: six groups, lr=0.0, eps=1e-15; learning rates of arguments/__init__.py:90-97),
and the three densification-statistic lines against a torch restatement of train.py:184-187 and
scene/gaussian_model.py:710-713."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import adam_oracle  # noqa: E402

# arguments/__init__.py:90-97, spatial_lr_scale = 1
GROUP_LRS = (("xyz", (3,), 0.00016), ("f_dc", (1, 3), 0.0025), ("f_rest", (15, 3), 0.0025 / 20.0),
             ("opacity", (1,), 0.05), ("scaling", (3,), 0.005), ("rotation", (4,), 0.001))


def make_params(N, seed):
    g = torch.Generator().manual_seed(seed)
    return {name: torch.randn((N,) + shape, generator=g) for name, shape, _ in GROUP_LRS}


def make_grads(N, seed, scale=1e-3):
    g = torch.Generator().manual_seed(seed)
    out = {}
    for name, shape, _ in GROUP_LRS:
        t = torch.randn((N,) + shape, generator=g) * scale
        t[torch.rand(N, generator=g) < 0.3] = 0.0          # culled Gaussians: exactly zero gradient
        out[name] = t
    return out


def _torch_sqrt(a):
    return torch.from_numpy(np.ascontiguousarray(a)).sqrt().numpy()


@pytest.mark.parametrize("sqrt", ["torch", "ieee"])
def test_oracle_equals_torch_adam_bit_for_bit(sqrt):
    """sqrt="torch": torch's CPU sqrt (MKL VML, one ulp off IEEE for 0.6 % of float32 inputs) is passed into the
    oracle -> everything is bit-identical.  sqrt="ieee" (numpy; what the GPU computes): the moments are still
    identical, the parameters within one ulp."""
    N, steps = 777, 6
    params = make_params(N, 0)
    tp = {k: torch.nn.Parameter(v.clone()) for k, v in params.items()}
    opt = torch.optim.Adam([{"params": [tp[name]], "lr": lr, "name": name} for name, _, lr in GROUP_LRS],
                           lr=0.0, eps=1e-15)
    op = {k: v.numpy().copy() for k, v in params.items()}
    om = {k: np.zeros_like(v) for k, v in op.items()}
    ov = {k: np.zeros_like(v) for k, v in op.items()}
    for t in range(1, steps + 1):
        grads = make_grads(N, 100 + t)
        lr_xyz = 0.00016 * (0.9 ** t)                         # the reference reschedules the xyz group every step
        opt.param_groups[0]["lr"] = lr_xyz
        for name, _, _ in GROUP_LRS:
            tp[name].grad = grads[name].clone()
        opt.step()
        for name, _, lr in GROUP_LRS:
            adam_oracle.adam_step(op[name], grads[name].numpy(), om[name], ov[name], t,
                                  lr_xyz if name == "xyz" else lr, form="cpu",
                                  sqrt=_torch_sqrt if sqrt == "torch" else np.sqrt)
        for name, _, _ in GROUP_LRS:
            st = opt.state[tp[name]]
            want = tp[name].detach().numpy()
            if sqrt == "torch":
                assert np.array_equal(op[name], want), (name, t)
            else:
                assert np.all(np.abs(op[name] - want) <= t * np.spacing(np.abs(want))), (name, t)
            assert np.array_equal(om[name], st["exp_avg"].numpy()), (name, t)
            assert np.array_equal(ov[name], st["exp_avg_sq"].numpy()), (name, t)


def test_cpu_and_gpu_forms_agree_to_rounding():
    """The GPU form (what the HIP kernel computes, bit-identical to torch on the GPU: tests/test_gpu_adam.py)
    differs from the pinned CPU form only in roundings."""
    N = 2000
    params, grads = make_params(N, 1), make_grads(N, 7)
    for name, _, lr in GROUP_LRS:
        a, b = params[name].numpy().copy(), params[name].numpy().copy()
        ma, va = np.zeros_like(a), np.zeros_like(a)
        mb, vb = np.zeros_like(a), np.zeros_like(a)
        for t in (1, 2, 3):
            adam_oracle.adam_step(a, grads[name].numpy(), ma, va, t, lr, form="cpu")
            adam_oracle.adam_step(b, grads[name].numpy(), mb, vb, t, lr, form="gpu")
        assert np.array_equal(ma, mb)
        assert np.all(np.abs(va - vb) <= 2 * np.spacing(va))
        assert np.abs(a - b).max() <= 4 * np.spacing(np.abs(a).max())


def test_stats_equal_torch_restatement():
    N, views = 500, 3
    g = torch.Generator().manual_seed(3)
    radii = torch.randint(0, 40, (views, N), generator=g, dtype=torch.int32)
    radii[torch.rand(views, N, generator=g) < 0.4] = 0
    vg = torch.randn(views, N, 3, generator=g)
    acc, den, mr = torch.rand(N, 1, generator=g), torch.randint(0, 5, (N, 1), generator=g).float(), \
        torch.randint(0, 30, (N,), generator=g).float()
    oacc, oden, omr = acc.numpy().copy().reshape(N), den.numpy().copy().reshape(N), mr.numpy().copy()
    for v in range(views):       # the reference's lines, one view per iteration
        vis = radii[v] > 0
        mr[vis] = torch.max(mr[vis], radii[v][vis])
        acc[vis] += torch.norm(vg[v][vis, :2], dim=-1, keepdim=True)
        den[vis] += 1
    adam_oracle.densify_stats(vg.numpy(), radii.numpy(), oacc, oden, omr)
    assert np.array_equal(omr, mr.numpy()) and np.array_equal(oden, den.numpy().reshape(N))
    assert np.allclose(oacc, acc.numpy().reshape(N), rtol=1e-6, atol=0)
