"""Drop-in ``diff_gaussian_rasterization`` package backed by the MI355X HIP rasterizer.

The reference discovers its accelerated back-end with
``from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer``
(reference gaussian_renderer/__init__.py:19-27).  Putting this repository's root on
``PYTHONPATH`` makes that import resolve here; nothing in the reference needs editing.
"""
import importlib as _importlib
import os as _os
import sys as _sys

_root = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
if _root not in _sys.path:
    _sys.path.insert(0, _root)

_impl = _importlib.import_module("gaussian-splatting_amd")

GaussianRasterizationSettings = _impl.GaussianRasterizationSettings
GaussianRasterizer = _impl.GaussianRasterizer
rasterize_gaussians = _impl.rasterize_gaussians
preprocess_and_bin = _impl.preprocess_and_bin
DEFAULT_T_EPS = _impl.DEFAULT_T_EPS

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "rasterize_gaussians"]
