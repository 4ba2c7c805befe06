"""Fused training loss on top of the C-ABI: (1 - lambda) * L1 + lambda * SSIM-term, value and gradient.

Host-side mirror of the reference's loss call site, train.py:141-146:

    Ll1  = l1_loss(image, gt_image)
    loss = (1.0 - opt.lambda_dssim) * Ll1 + opt.lambda_dssim * (1.0 - ssim_loss(image[None], gt_image[None]))

``ssim_loss`` already is ``1 - SSIM`` (utils/loss_utils.py:193-205), so the reference's second term is
``+lambda * SSIM``.  ``reference_sign=True`` (default) reproduces exactly that; ``reference_sign=False``
gives the D-SSIM loss of the 3DGS paper, ``(1 - lambda) * L1 + lambda * (1 - SSIM)``.

Three kernels (forward maps + partial sums, final reduction, backward blur) replace the reference's
~30 autograd nodes and ten 11x11 grouped convolutions.  No CPU fallback.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


class _L1SSIM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, image, gt, w_l1, w_ssim):
        lib = _lib.load()
        if not image.is_cuda or not gt.is_cuda:
            raise RuntimeError("l1_ssim_loss needs GPU tensors; this back-end has no CPU path")
        if image.shape != gt.shape or image.dim() != 3:
            raise ValueError(f"image and gt must both be (C,H,W); got {tuple(image.shape)} and {tuple(gt.shape)}")
        dev = image.device
        x = image.detach().to(torch.float32).contiguous()
        y = gt.detach().to(device=dev, dtype=torch.float32).contiguous()
        Cc, H, W = x.shape
        with torch.cuda.device(dev):
            ws_bytes = lib.gs_l1_ssim_workspace_bytes(Cc, H, W)
            ws = torch.empty((max(ws_bytes, 16),), dtype=torch.uint8, device=dev)
            out = torch.empty((3,), dtype=torch.float32, device=dev)
            grad = torch.empty_like(x) if image.requires_grad else None
            _lib.check(lib.gs_l1_ssim(Cc, H, W, _ptr(x), _ptr(y), float(w_l1), float(w_ssim), _ptr(out), _ptr(grad),
                                      _ptr(ws), ws_bytes, C.c_void_p(torch.cuda.current_stream().cuda_stream)),
                       "gs_l1_ssim")
        ctx.grad = grad
        ctx.mark_non_differentiable(out)
        return out[0].clone(), out

    @staticmethod
    def backward(ctx, g_loss, _g_parts):
        if ctx.grad is None:
            return None, None, None, None
        return ctx.grad * g_loss, None, None, None


def l1_ssim_loss(image: torch.Tensor, gt: torch.Tensor, lambda_dssim: float = 0.2, reference_sign: bool = True,
                 return_parts: bool = False):
    """Scalar training loss for (C,H,W) ``image`` against ``gt``; differentiable w.r.t. ``image``.

    return_parts: also return a (3,) tensor (loss without any constant term, L1, mean SSIM) for logging.
    """
    lam = float(lambda_dssim)
    if reference_sign:
        loss, parts = _L1SSIM.apply(image, gt, 1.0 - lam, lam)
    else:
        loss, parts = _L1SSIM.apply(image, gt, 1.0 - lam, -lam)
        loss = loss + lam
    return (loss, parts) if return_parts else loss
