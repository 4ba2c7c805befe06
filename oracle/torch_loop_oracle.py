"""PyTorch-CPU restatement of the reference's per-Gaussian splat loop (TEST / BASELINE INFRASTRUCTURE ONLY: imported
by tests/ and by bench.py's cpu_baseline leg, never by the product path).

SURVEY.md 8d (ii) asks for the build's own PyTorch-CPU restatement of the loop next to the C port for config 1
(10 k Gaussians, SH degree 0, 200x200): it shows what the reference's *form* of the algorithm costs -- ~15 small
torch ops and four host-synchronising comparisons per Gaussian, autograd recording ~40 nodes each -- as opposed to
the arithmetic itself, which oracle/gs_oracle.c runs 400x faster on the same cores.

Follows gaussian_renderer/__init__.py of the reference: projection :116-133, compute_cov3D :210-264,
project_cov3D_to_2D :291-334 (focal from projmatrix, no view rotation in J, +0.3), far-to-near order :151-159,
splat loop :366-440 (culls :368-381, 2x2 inverse :384-389, integer box :393-402, weight :417-426, compositing
:429-440), SH degree 0 colours utils/sh_utils.py:9,64 + clamp :206.  Pinned by tests/test_torch_loop_oracle.py
against the goldens recorded from the reference.  Degree 0 / scales+rotations only (config 1's inputs).
"""
import torch

C0 = 0.28209479177387814


def _geometry(means3D, scales, rotations, viewmatrix, projmatrix, H, W):
    N = means3D.shape[0]
    ph = torch.cat([means3D, torch.ones(N, 1)], dim=1)
    cam = (viewmatrix @ ph.T).T
    clip = (projmatrix @ ph.T).T
    ndc = clip[:, :3] / (clip[:, 3:4] + 1e-7)
    sx = ((ndc[:, 0] + 1.0) * W - 1.0) * 0.5
    sy = ((ndc[:, 1] + 1.0) * H - 1.0) * 0.5
    r, x, y, z = rotations[:, 0], rotations[:, 1], rotations[:, 2], rotations[:, 3]
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                     2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                     2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=1).reshape(N, 3, 3)
    M = R @ torch.diag_embed(scales)
    cov3D = M @ M.transpose(1, 2)
    fx, fy = projmatrix[0, 0] * W / 2.0, projmatrix[1, 1] * H / 2.0
    cx, cy, cz = cam[:, 0], cam[:, 1], cam[:, 2]
    J = torch.zeros(N, 2, 3)
    J[:, 0, 0] = fx / (cz + 1e-7)
    J[:, 0, 2] = -cx * fx / (cz * cz + 1e-7)
    J[:, 1, 1] = fy / (cz + 1e-7)
    J[:, 1, 2] = -cy * fy / (cz * cz + 1e-7)
    cov2D = J @ cov3D @ J.transpose(1, 2)
    cov2D = cov2D + 0.3 * torch.eye(2)
    return torch.stack([sx, sy], dim=1), cam[:, 2], cov2D


def render(means3D, shs_dc, opacities, scales, rotations, viewmatrix, projmatrix, bg, H, W):
    """shs_dc (N,3): the degree-0 coefficients.  Returns (image (3,H,W), depth (1,H,W), alpha (1,H,W))."""
    colors = torch.clamp(C0 * shs_dc + 0.5, 0.0, 1.0)
    xy, depths, cov2D = _geometry(means3D, scales, rotations, viewmatrix, projmatrix, H, W)
    order = torch.argsort(depths, descending=True, stable=True)
    image = bg.reshape(3, 1, 1).expand(3, H, W).clone()
    depth = torch.zeros(1, H, W)
    acc = torch.zeros(1, H, W)
    for i in order.tolist():                      # the reference's hot loop, one Gaussian at a time
        mx, my = xy[i, 0], xy[i, 1]
        if mx < -50 or mx > W + 50 or my < -50 or my > H + 50:      # host syncs, as in the reference
            continue
        c = cov2D[i]
        det = c[0, 0] * c[1, 1] - c[0, 1] * c[1, 0]
        if det < 1e-7:
            continue
        inv = torch.stack([c[1, 1], -c[0, 1], -c[1, 0], c[0, 0]]).reshape(2, 2) / det
        rx, ry = 3.0 * torch.sqrt(c[0, 0]), 3.0 * torch.sqrt(c[1, 1])
        x0, x1 = int(max(0, mx - rx)), int(min(W, mx + rx + 1))
        y0, y1 = int(max(0, my - ry)), int(min(H, my + ry + 1))
        if x1 <= x0 or y1 <= y0:
            continue
        ys, xs = torch.meshgrid(torch.arange(y0, y1, dtype=torch.float32), torch.arange(x0, x1, dtype=torch.float32),
                                indexing="ij")
        dx, dy = xs - mx, ys - my
        power = -0.5 * (inv[0, 0] * dx * dx + (inv[0, 1] + inv[1, 0]) * dx * dy + inv[1, 1] * dy * dy)
        w = torch.exp(power) * opacities[i, 0]
        a = w * (1.0 - acc[0, y0:y1, x0:x1])
        image[:, y0:y1, x0:x1] = image[:, y0:y1, x0:x1] + a.unsqueeze(0) * colors[i].reshape(3, 1, 1)
        acc[0, y0:y1, x0:x1] = acc[0, y0:y1, x0:x1] + a
        depth[0, y0:y1, x0:x1] = depth[0, y0:y1, x0:x1] + a * depths[i]
    return image, depth, acc
