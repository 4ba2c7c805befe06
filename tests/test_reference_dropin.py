"""CPU-only, build container only: the UNMODIFIED reference discovers this back-end.

Skipped wherever /root/reference is absent (the GPU box); nothing from it is copied."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"

pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "gaussian_renderer")),
                                reason="reference checkout not present")

PROBE = r"""
import sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[2])
import gaussian_renderer as gr
import diff_gaussian_rasterization as ours
assert gr.CUDA_RASTERIZER_AVAILABLE is True, "reference did not pick the back-end up"
assert gr.CUDAGaussianRasterizer is ours.GaussianRasterizer
assert gr.CUDAGaussianRasterizationSettings is ours.GaussianRasterizationSettings
# the adapter's settings call (gaussian_renderer/__init__.py:563-583) constructs without the TypeError retry
import torch
s = gr.CUDAGaussianRasterizationSettings(image_height=8, image_width=8, tanfovx=1.0, tanfovy=1.0, bg=torch.zeros(3),
    scale_modifier=1.0, viewmatrix=torch.eye(4), projmatrix=torch.eye(4), sh_degree=0, campos=torch.zeros(3),
    prefiltered=False, debug=False, antialiasing=False)
r = gr.CUDAGaussianRasterizer(raster_settings=s)
print("OK")
"""


def test_reference_imports_our_backend():
    out = subprocess.run([sys.executable, "-c", PROBE, ROOT, REF], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "OK" in out.stdout, out.stderr[-2000:]
