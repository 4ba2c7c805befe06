"""MI355X-native differentiable Gaussian rasterizer (hand-written HIP for gfx950 behind a C-ABI).

The directory name is not a Python identifier; import it through the drop-in package
``diff_gaussian_rasterization`` (the name the reference imports) or with
``importlib.import_module("gaussian-splatting_amd")``.
"""
from .rasterizer import (DEFAULT_T_EPS, BinningState, GaussianRasterizationSettings, GaussianRasterizer,
                         preprocess_and_bin, rasterize_gaussians)
from .loss import l1_ssim_loss
from .optimizer import FusedAdam
from .ply_io import load_ply, save_ply
from .knn import distCUDA2
from . import _lib

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "rasterize_gaussians",
           "preprocess_and_bin", "BinningState", "DEFAULT_T_EPS", "l1_ssim_loss", "FusedAdam", "load_ply", "save_ply", "distCUDA2"]
__version__ = "0.1.0"

# Fail loudly at import: the reference swallows import errors (gaussian_renderer/__init__.py:26-27)
# and silently falls back to its slow PyTorch loop, so say what happened before raising.
try:
    _lib.load()
except Exception as _e:  # noqa: BLE001
    import sys as _sys
    print(f"[gaussian-splatting_amd] FATAL: HIP back-end unavailable: {_e}", file=_sys.stderr, flush=True)
    raise
