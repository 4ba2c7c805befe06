#!/usr/bin/env python3
"""Golden vectors for the training loss, recorded by running the REFERENCE's own
functions.  Runs ONLY in the build container (needs /root/reference, read-only); the GPU box never runs
this.  Nothing from the reference is copied: it is imported, called, and its outputs are recorded.

    python tests/golden/gen_golden_loss.py        # writes tests/golden/loss/*.npz

Reference entry points used (file:line in /root/reference):
  utils/loss_utils.py:10   l1_loss
  utils/loss_utils.py:193  ssim_loss (-> calculate_ssim :118 -> ssim :74, create_window :56)
  train.py:141-146         loss = (1 - lambda) * L1 + lambda * (1 - ssim_loss(img[None], gt[None]))
                           (sic: 1 - ssim_loss = SSIM itself; recorded as the reference computes it)
Each file: img, gt (C,H,W) float32, lambda, l1, ssim (mean SSIM), loss, and d loss / d img from the
reference's autograd.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")
from utils.loss_utils import gaussian, l1_loss, ssim_loss  # noqa: E402  (the reference)

torch.set_num_threads(8)
# the reference's 1-D window, bit for bit (the kernels and the oracle embed these 11 floats)
np.savez(os.path.join(HERE, "loss", "_window_1d.npz"), g=gaussian(11, 1.5).numpy())
CASES = [  # name, C, H, W, seed, lambda, kind
    ("rand_3x37x53", 3, 37, 53, 0, 0.2, "rand"),
    ("smooth_3x64x64", 3, 64, 64, 1, 0.2, "smooth"),
    ("tiny_1x7x9", 1, 7, 9, 2, 0.2, "rand"),            # smaller than the 11x11 window
    ("close_3x48x80", 3, 48, 80, 3, 0.5, "close"),       # prediction near the target, some exact ties
    ("photo_3x200x200", 3, 200, 200, 4, 0.2, "smooth"),
]


def make(kind, C, H, W, g):
    if kind == "rand":
        return torch.rand(C, H, W, generator=g), torch.rand(C, H, W, generator=g)
    yy, xx = torch.meshgrid(torch.linspace(0, 1, H), torch.linspace(0, 1, W), indexing="ij")
    base = torch.stack([0.5 + 0.4 * torch.sin(6.0 * xx + 3.0 * yy + c) * torch.cos(4.0 * yy - c) for c in range(C)])
    if kind == "smooth":
        return (base + 0.1 * torch.randn(C, H, W, generator=g)).clamp(0, 1), base.clamp(0, 1)
    img = (base + 0.01 * torch.randn(C, H, W, generator=g)).clamp(0, 1)
    img[:, ::5, ::7] = base.clamp(0, 1)[:, ::5, ::7]     # exact ties: d|x|/dx = 0 there
    return img, base.clamp(0, 1)


for name, C, H, W, seed, lam, kind in CASES:
    g = torch.Generator().manual_seed(seed)
    img, gt = make(kind, C, H, W, g)
    img = img.float().contiguous().requires_grad_(True)
    gt = gt.float().contiguous()
    l1 = l1_loss(img, gt)
    sl = ssim_loss(img.unsqueeze(0), gt.unsqueeze(0))
    loss = (1.0 - lam) * l1 + lam * (1.0 - sl)            # train.py:143-145
    loss.backward()
    np.savez_compressed(os.path.join(HERE, "loss", name + ".npz"), img=img.detach().numpy(), gt=gt.numpy(),
                        lam=np.float32(lam), l1=np.float32(l1.item()), ssim=np.float32(1.0 - sl.item()),
                        loss=np.float32(loss.item()), grad=img.grad.numpy())
    print(name, float(l1), float(1.0 - sl), float(loss), float(img.grad.abs().max()))
