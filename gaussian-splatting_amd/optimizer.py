"""Fused optimizer step of the training loop: host side of gs_adam_densify_step.

``FusedAdam`` is a ``torch.optim.Adam`` whose ``step()`` runs as ONE HIP launch over all parameter
groups and, when the rasterizer's outputs are passed along, also does the two statistic updates
the reference's trainer performs around it:

    reference                                                         here
    gaussians.optimizer = torch.optim.Adam(l, lr=0.0, eps=1e-15)      FusedAdam(l, lr=0.0, eps=1e-15)
      (scene/gaussian_model.py:297)
    max_radii2D[vis] = max(max_radii2D[vis], radii[vis])              optimizer.step(viewspace_grad=..., radii=...,
      (train.py:184-187)                                                  xyz_gradient_accum=..., denom=...,
    add_densification_stats(viewspace_points, vis)                        max_radii2D=...)
      (scene/gaussian_model.py:701-713)
    optimizer.step()   (train.py:217)

It keeps torch's ``param_groups`` and ``state`` (``step``, ``exp_avg``, ``exp_avg_sq`` per parameter), so the
reference's learning-rate schedule (:307-314), densification surgery on the optimizer state
(:461-530), ``state_dict()`` / ``load_state_dict()`` and ``zero_grad()`` work unchanged.
No CPU path: parameters must be float32 on the GPU.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from .profiling import stage as _stage

MAX_GROUPS = 8   # AD_MAX_GROUPS of csrc/optimizer.hip


def _rows_view(g: torch.Tensor, shape):
    """grad as a (rows, row_len) matrix with contiguous rows -> (tensor to keep alive, row_stride)."""
    row_len = 1
    for d in shape[1:]:
        row_len *= d
    if g.dim() >= 1 and g.shape[0] > 1 and g[0].is_contiguous() and g.stride(0) >= row_len:
        return g, g.stride(0)
    g = g.contiguous()
    return g, row_len


class FusedAdam(torch.optim.Adam):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False, **kw):
        if weight_decay != 0 or amsgrad or kw.get("maximize", False):
            raise NotImplementedError("FusedAdam implements the reference's Adam: no weight_decay / amsgrad / maximize")
        for k in ("foreach", "capturable", "fused", "differentiable"):
            kw.pop(k, None)
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False, foreach=False)

    @torch.no_grad()
    def update_densify_stats(self, viewspace_grad=None, radii=None, xyz_gradient_accum=None, denom=None,
                             max_radii2D=None):
        """The statistics of train.py:184-193 / scene/gaussian_model.py:701-713 alone, no parameter update
        (gs_adam_densify_step with zero groups: one small launch).

        This is the call for iterations on which the reference densifies, prunes or resets between the
        statistics and the step (train.py:196-214): densify() consumes, resizes and zeroes the accumulators,
        so they must be up to date BEFORE it -- and after it radii / viewspace_grad have the old N while the
        accumulators have the new one, so they cannot be folded into step() on such an iteration."""
        if radii is None:
            raise ValueError("update_densify_stats needs radii (the rasterizer's second output)")
        self._launch([], self._check_stats(viewspace_grad, radii, xyz_gradient_accum, denom, max_radii2D), radii.device)

    @staticmethod
    def _check_stats(viewspace_grad, radii, xyz_gradient_accum, denom, max_radii2D):
        if not radii.is_cuda or radii.dtype != torch.int32:
            raise RuntimeError("radii must be an int32 tensor on the GPU (the rasterizer's second output)")
        radii = radii.contiguous()
        views = 1 if radii.dim() == 1 else radii.shape[0]
        n = radii.shape[-1]
        if viewspace_grad is not None:
            viewspace_grad = viewspace_grad.contiguous()
            if viewspace_grad.numel() != views * n * 3 or viewspace_grad.dtype != torch.float32:
                raise ValueError("viewspace_grad must be float32 with shape (N,3) or (views,N,3) matching radii")
        for name, t_ in (("xyz_gradient_accum", xyz_gradient_accum), ("denom", denom), ("max_radii2D", max_radii2D)):
            if t_ is not None and (t_.numel() != n or t_.dtype != torch.float32 or not t_.is_contiguous()
                                   or not t_.is_cuda):
                raise ValueError(f"{name} must be a contiguous float32 GPU tensor with N = {n} elements (radii has "
                                 f"{n}: were the Gaussians densified or pruned since the render?)")
        return (n, views, viewspace_grad, radii, xyz_gradient_accum, denom, max_radii2D)

    @torch.no_grad()
    def step(self, closure=None, *, viewspace_grad=None, radii=None, xyz_gradient_accum=None, denom=None,
             max_radii2D=None):
        """One Adam update of every parameter that has a .grad (others are skipped, as in torch).

        Optional densification statistics of this step (all on the GPU): ``radii`` (N,) or (views,N) int32,
        ``viewspace_grad`` (N,3) or (views,N,3) = means2D.grad, and the (N,) or (N,1) float32 accumulators
        to update in place.  They ride along in the same launch -- valid only when nothing resizes or resets
        the accumulators between the backward and this step (no densify / prune on this iteration); otherwise
        call update_densify_stats() where the reference updates them and step() without these arguments."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        # checked first: a refused call must leave the step counters untouched
        stats = None if radii is None else self._check_stats(viewspace_grad, radii, xyz_gradient_accum, denom, max_radii2D)
        buckets = {}     # (beta1, beta2, eps, step) -> list of (GsAdamGroup fields, keep-alive tensors)
        dev = None
        for group in self.param_groups:
            if group.get("weight_decay", 0) != 0 or group.get("amsgrad", False) or group.get("maximize", False):
                raise NotImplementedError("FusedAdam: weight_decay / amsgrad / maximize are not implemented")
            beta1, beta2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                    raise RuntimeError("FusedAdam needs contiguous float32 parameters on the GPU (no CPU path)")
                g = p.grad
                if g.is_sparse or g.dtype != torch.float32 or g.device != p.device:
                    raise RuntimeError("FusedAdam needs dense float32 gradients on the parameter's device")
                if dev is None:
                    dev = p.device
                elif dev != p.device:
                    raise RuntimeError("FusedAdam: all parameters must live on one device")
                state = self.state[p]
                if len(state) == 0:       # torch/optim/adam.py: lazy state initialisation
                    state["step"] = torch.tensor(0.0, dtype=torch.float32)
                    state["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    state["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                m, v = state["exp_avg"], state["exp_avg_sq"]
                if not (m.is_contiguous() and v.is_contiguous()) or m.shape != p.shape or v.shape != p.shape:
                    raise RuntimeError("FusedAdam: optimizer state does not match its parameter")
                state["step"] += 1
                t = int(state["step"].item()) if torch.is_tensor(state["step"]) else int(state["step"])
                rows = p.shape[0] if p.dim() >= 1 else 1
                row_len = p.numel() // rows if rows else 1
                gk, stride = _rows_view(g, p.shape if p.dim() >= 1 else (1,))
                lr = group["lr"]
                lr = float(lr.item()) if torch.is_tensor(lr) else float(lr)
                buckets.setdefault((float(beta1), float(beta2), float(group["eps"]), t), []).append(
                    ((p.data_ptr(), gk.data_ptr(), m.data_ptr(), v.data_ptr(), rows, stride, max(row_len, 1), lr),
                     (p, gk, m, v)))
        if stats is not None:
            dev = dev or radii.device
        if dev is None:
            return loss
        self._launch(buckets, stats, dev)
        return loss

    def _launch(self, buckets, stats, dev):
        lib = _lib.load()
        calls = []
        for key, items in (buckets.items() if buckets else []):
            for s in range(0, len(items), MAX_GROUPS):
                calls.append((key, items[s:s + MAX_GROUPS]))
        if not calls:
            calls.append(((0.9, 0.999, 1e-8, 1), []))
        ptr = lambda t_: None if t_ is None else C.c_void_p(t_.data_ptr())  # noqa: E731
        with torch.cuda.device(dev), _stage("adam_densify"):
            st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            for ci, ((beta1, beta2, eps, t), items) in enumerate(calls):
                arr = (_lib.GsAdamGroup * max(len(items), 1))()
                for k, (f, _) in enumerate(items):
                    arr[k] = _lib.GsAdamGroup(f[0], f[1], f[2], f[3], f[4], f[5], f[6], 0, f[7])
                if stats is not None and ci == len(calls) - 1:
                    n, views, vg, rd, acc, den, mr = stats
                    rc = lib.gs_adam_densify_step(len(items), arr, beta1, beta2, eps, t, n, views, ptr(vg), ptr(rd),
                                                  ptr(acc), ptr(den), ptr(mr), st)
                else:
                    rc = lib.gs_adam_densify_step(len(items), arr, beta1, beta2, eps, t, 0, 0, None, None, None, None,
                                                  None, st)
                _lib.check(rc, "gs_adam_densify_step")
