"""GPU: the rows chained the way the reference's trainer chains them (train.py:129-218) -- raw parameters ->
activations (scene/gaussian_model.py:104-142) -> rasterizer -> L1 + SSIM loss -> backward -> Adam with the reference's
six groups -> densification statistics -- run twice over several views:
  (A) this repository's fused pieces: GaussianRasterizer + l1_ssim_loss + FusedAdam.step(viewspace_grad=..., radii=...)
  (B) the same rasterizer with the loss and the optimizer as the reference composes them from torch ops
      (l1 = mean|x-y|, SSIM from five grouped conv2d with the 11x11 window of utils/loss_utils.py:38-118, restated here;
      torch.optim.Adam; the three statistic lines).
Both must fit the target views (loss falls), stay within a small distance of each other after 25 iterations (they
differ only by the loss kernels' float32 rounding, 2e-6 in value), and accumulate identical statistics counts.
"""
import importlib
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

LRS = {"xyz": 0.00016 * 5.0, "f_dc": 0.0025, "f_rest": 0.0025 / 20.0, "opacity": 0.05, "scaling": 0.005, "rotation": 0.001}


def _window(dev):
    g = torch.tensor([math.exp(-((x - 5) ** 2) / (2 * 1.5 ** 2)) for x in range(11)])
    g = (g / g.sum()).unsqueeze(1)
    return (g @ g.t()).float().unsqueeze(0).unsqueeze(0).expand(3, 1, 11, 11).contiguous().to(dev)


def _torch_loss(img, gt, lam, win, reference_sign):
    """reference_sign: (1 - lam) * l1_loss + lam * (1 - ssim_loss) with ssim_loss = 1 - SSIM, i.e. + lam * SSIM
    (train.py:141-146, utils/loss_utils.py); otherwise the paper's (1 - lam) * L1 + lam * (1 - SSIM)."""
    l1 = (img - gt).abs().mean()
    x, y = img.unsqueeze(0), gt.unsqueeze(0)
    mu1, mu2 = F.conv2d(x, win, padding=5, groups=3), F.conv2d(y, win, padding=5, groups=3)
    s11 = F.conv2d(x * x, win, padding=5, groups=3) - mu1 * mu1
    s22 = F.conv2d(y * y, win, padding=5, groups=3) - mu2 * mu2
    s12 = F.conv2d(x * y, win, padding=5, groups=3) - mu1 * mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    ssim = (((2 * mu1 * mu2 + C1) * (2 * s12 + C2)) / ((mu1 * mu1 + mu2 * mu2 + C1) * (s11 + s22 + C2))).mean()
    return (1.0 - lam) * l1 + (lam * ssim if reference_sign else lam * (1.0 - ssim))


def _run(fused, impl, syn, reference_sign, iters=25):
    dev = torch.device("cuda:0")
    N, K, H, W = 4000, 16, 96, 128
    target = syn.make_scene(N, K, H, W, seed=21)
    views = [syn.jitter_view(target, v, sigma=0.1) for v in range(5)]

    def settings(v):
        return impl.GaussianRasterizationSettings(
            image_height=H, image_width=W, tanfovx=v.tanfovx, tanfovy=v.tanfovy, bg=v.bg.to(dev), scale_modifier=1.0,
            viewmatrix=v.viewmatrix.to(dev), projmatrix=v.projmatrix.to(dev), sh_degree=v.sh_degree,
            campos=v.campos.to(dev), prefiltered=False, debug=False)

    sets = [settings(v) for v in views]
    with torch.no_grad():
        gts = [impl.GaussianRasterizer(s)(means3D=target.means3D.to(dev), shs=target.shs.to(dev),
                                          opacities=target.opacities.to(dev), scales=target.scales.to(dev),
                                          rotations=target.rotations.to(dev))[0].clamp(0, 1) for s in sets]
    # the model starts from a perturbed copy, stored as the reference stores it (raw, pre-activation)
    g = torch.Generator().manual_seed(5)
    P = torch.nn.Parameter
    raw = {"xyz": P((target.means3D + 0.01 * torch.randn(N, 3, generator=g)).to(dev)),
           "f_dc": P((target.shs[:, :1] + 0.2 * torch.randn(N, 1, 3, generator=g)).to(dev).contiguous()),
           "f_rest": P((target.shs[:, 1:] * 0.5).to(dev).contiguous()),
           "opacity": P(torch.logit(target.opacities.clamp(0.02, 0.98)).to(dev)),
           "scaling": P(torch.log(target.scales * 1.2).to(dev)),
           "rotation": P((target.rotations * 2.0).to(dev))}
    groups = [{"params": [raw[k]], "lr": LRS[k], "name": k} for k in LRS]
    opt = (impl.FusedAdam if fused else torch.optim.Adam)(groups, lr=0.0, eps=1e-15)
    acc, den, mr = torch.zeros(N, 1, device=dev), torch.zeros(N, 1, device=dev), torch.zeros(N, device=dev)
    win = _window(dev)
    losses = []
    for it in range(iters):
        v = it % len(sets)
        m2d = torch.zeros(N, 3, device=dev, requires_grad=True)
        image, radii, _, _ = impl.GaussianRasterizer(sets[v])(
            means3D=raw["xyz"], means2D=m2d, shs=torch.cat((raw["f_dc"], raw["f_rest"]), dim=1),
            opacities=torch.sigmoid(raw["opacity"]), scales=torch.exp(raw["scaling"]),
            rotations=F.normalize(raw["rotation"]))
        loss = impl.l1_ssim_loss(image, gts[v], lambda_dssim=0.2, reference_sign=reference_sign) if fused else \
            _torch_loss(image, gts[v], 0.2, win, reference_sign)
        loss.backward()
        if fused:
            opt.step(viewspace_grad=m2d.grad, radii=radii, xyz_gradient_accum=acc, denom=den, max_radii2D=mr)
        else:
            vis = radii > 0                                        # train.py:184-193, scene/gaussian_model.py:710-713
            mr[vis] = torch.max(mr[vis], radii[vis].float())
            acc[vis] += torch.norm(m2d.grad[vis, :2], dim=-1, keepdim=True)
            den[vis] += 1
            opt.step()
        opt.zero_grad(set_to_none=True)
        losses.append((float(loss.detach()), float((image.detach() - gts[v]).abs().mean())))
    return losses, {k: v.detach().clone() for k, v in raw.items()}, (acc, den, mr)


@pytest.mark.parametrize("reference_sign", [True, False])
def test_fused_training_loop_tracks_the_torch_composed_one(reference_sign):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    impl = importlib.import_module("gaussian-splatting_amd")
    syn = importlib.import_module("gaussian-splatting_amd.synthetic")
    la, pa, (acc_a, den_a, mr_a) = _run(True, impl, syn, reference_sign)
    lb, pb, (acc_b, den_b, mr_b) = _run(False, impl, syn, reference_sign)
    assert abs(la[0][0] - lb[0][0]) <= 5e-6
    for a, b in zip(la, lb):
        assert abs(a[0] - b[0]) <= 2e-3 and abs(a[1] - b[1]) <= 2e-3, (a, b)
    assert torch.equal(den_a, den_b) and torch.equal(mr_a, mr_b)
    assert (acc_a - acc_b).abs().max().item() <= 1e-2 * acc_b.abs().max().item()
    for k in pa:
        d = (pa[k] - pb[k]).abs().max().item()
        assert d <= 25 * LRS[k] * 0.5, (k, d)          # far inside the 25 * lr an Adam step sequence can move
    # and the loop really trains: with the paper's sign the same five views are closer to their targets after 20
    # more iterations.  (With the reference's own sign -- "+ lam * SSIM", train.py:143-145 -- gradient descent lowers
    # the SSIM and the views drift away from the targets; both loops reproduce that faithfully, see above.)
    first, last = sum(x[1] for x in la[:5]) / 5, sum(x[1] for x in la[-5:]) / 5
    if not reference_sign:
        assert last < 0.9 * first, (first, last)
