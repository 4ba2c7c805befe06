"""Worker of tests/test_gpu_exchange.py (not collected by pytest): one rank of a SplatExchange step.

Launched `world` times with RANK / WORLD_SIZE / MASTER_PORT in the environment; every rank uses cuda:0
(the GPU box has one GPU) and the gloo backend (RCCL refuses two ranks on one device), so the
collectives are staged through the host -- the step's logic and kernels are the ones the RCCL run uses.
Checks against a single-process render of the whole scene on the same GPU:
  * own view's image / depth / alpha bit-identical, radii of every view identical,
  * gradients of the local shard == sum over the views of the single-GPU gradients (fp32 tolerance:
    the render_bwd atomics add in a different order),
  * per-view means2D gradients.
"""
import importlib
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

TOL_GRAD_REL = 1e-4      # of max|grad| per tensor (summation order only)


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    impl = importlib.import_module("gaussian-splatting_amd")
    syn = importlib.import_module("gaussian-splatting_amd.synthetic")
    par = importlib.import_module("gaussian-splatting_amd.parallel")
    N, K, H, W = 30000, 16, 200, 312
    base = syn.make_scene(N, K, H, W, seed=3)
    # uneven shards: rank r owns [bounds[r], bounds[r+1])
    bounds = [0] + [int(N * (r + 1) / world * (0.8 if r + 1 < world else 1.0)) for r in range(world)]
    if os.environ.get("EXCHANGE_VARIANT") == "empty_shard":       # the last rank owns nothing (after heavy pruning, say)
        bounds = [0] + [N] * world
        bounds[1:world] = [int(N * (r + 1) / (world - 1)) if world > 2 else N for r in range(world - 1)]
    lo, hi = bounds[rank], bounds[rank + 1]
    names = ("means3D", "shs", "opacities", "scales", "rotations")
    views = [syn.jitter_view(base, v) for v in range(world)]

    def settings(v):
        return impl.GaussianRasterizationSettings(
            image_height=H, image_width=W, tanfovx=v.tanfovx, tanfovy=v.tanfovy, bg=v.bg.to(dev), scale_modifier=1.0,
            viewmatrix=v.viewmatrix.to(dev), projmatrix=v.projmatrix.to(dev), sh_degree=v.sh_degree,
            campos=v.campos.to(dev), prefiltered=False, debug=False)

    sets = [settings(v) for v in views]
    gen = torch.Generator().manual_seed(100)
    Gs = [(torch.rand(3, H, W, generator=gen) * 2 - 1).to(dev) for _ in range(world)]
    Gd = [(torch.rand(1, H, W, generator=gen) * 0.1).to(dev) for _ in range(world)]

    if os.environ.get("EXCHANGE_VARIANT") == "precomp":
        return precomp_variant(impl, par, base, sets, Gs, lo, hi, N, rank, world, dev)

    # ---- sharded step
    local = {k: getattr(base, k)[lo:hi].to(dev).clone().requires_grad_(True) for k in names}
    m2d = torch.zeros((world, hi - lo, 3), device=dev, requires_grad=True)
    ex = par.SplatExchange(hi - lo, device=dev)
    assert ex.counts == [bounds[r + 1] - bounds[r] for r in range(world)] and ex.offset == lo
    image, radii, depth, alpha = ex.rasterize(sets, local["means3D"], means2D=m2d, shs=local["shs"],
                                              opacities=local["opacities"], scales=local["scales"],
                                              rotations=local["rotations"])
    ((image * Gs[rank]).sum() + (depth * Gd[rank]).sum()).backward()

    # ---- single-process reference: every view of the whole scene, gradients accumulated in view order
    full = {k: getattr(base, k).to(dev).clone().requires_grad_(True) for k in names}
    ref_m2d, ref_radii, ref_out = [], [], None
    for v in range(world):
        mv = torch.zeros((N, 3), device=dev, requires_grad=True)
        out = impl.GaussianRasterizer(sets[v])(means3D=full["means3D"], means2D=mv, shs=full["shs"],
                                               opacities=full["opacities"], scales=full["scales"],
                                               rotations=full["rotations"])
        ((out[0] * Gs[v]).sum() + (out[2] * Gd[v]).sum()).backward()
        ref_m2d.append(mv.grad[lo:hi])
        ref_radii.append(out[1][lo:hi])
        if v == rank:
            ref_out = [t.detach() for t in out]
    assert torch.equal(image, ref_out[0]), "image differs from the single-GPU render"
    assert torch.equal(depth, ref_out[2]) and torch.equal(alpha, ref_out[3])
    assert torch.equal(radii, torch.stack(ref_radii))
    worst = {}
    if hi == lo:                                   # empty shard: empty gradients of the right shapes, nothing else
        for k in names:
            assert local[k].grad is not None and local[k].grad.shape == local[k].shape
        assert m2d.grad.shape == (world, 0, 3)
        torch.cuda.synchronize()
        print(f"rank {rank}: OK (empty shard)", flush=True)
        dist.barrier()
        dist.destroy_process_group()
        return
    for k in names:
        ref = full[k].grad[lo:hi]
        err = (local[k].grad - ref).abs().max().item()
        worst[k] = err / max(ref.abs().max().item(), 1e-30)
        assert worst[k] <= TOL_GRAD_REL, (k, worst[k])
    ref = torch.stack(ref_m2d)
    e = (m2d.grad - ref).abs().max().item() / max(ref.abs().max().item(), 1e-30)
    assert e <= TOL_GRAD_REL, ("means2D", e)
    assert ref.abs().max().item() > 0
    torch.cuda.synchronize()
    print(f"rank {rank}: OK", {k: f"{v:.1e}" for k, v in worst.items()}, f"means2D {e:.1e}", flush=True)
    dist.barrier()
    dist.destroy_process_group()


def precomp_variant(impl, par, base, sets, Gs, lo, hi, N, rank, world, dev):
    """colors_precomp + cov3D_precomp inputs, no means2D: the other argument combination of the rasterizer."""
    g = torch.Generator().manual_seed(77)
    colors = torch.rand(N, 3, generator=g)
    Rm = torch.randn(N, 3, 3, generator=g) * base.scales.mean().item()
    S = Rm @ Rm.transpose(1, 2) + 1e-6 * torch.eye(3)
    cov6 = torch.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], 1).contiguous()
    full = {"means3D": base.means3D, "colors": colors, "opacities": base.opacities, "cov": cov6}
    loc = {k: v[lo:hi].to(dev).clone().requires_grad_(True) for k, v in full.items()}
    ex = par.SplatExchange(hi - lo, device=dev)
    image, radii, depth, alpha = ex.rasterize(sets, loc["means3D"], colors_precomp=loc["colors"],
                                              opacities=loc["opacities"], cov3D_precomp=loc["cov"])
    (image * Gs[rank]).sum().backward()
    ref = {k: v.to(dev).clone().requires_grad_(True) for k, v in full.items()}
    mine = None
    for v in range(world):
        out = impl.GaussianRasterizer(sets[v])(means3D=ref["means3D"], colors_precomp=ref["colors"],
                                               opacities=ref["opacities"], cov3D_precomp=ref["cov"])
        (out[0] * Gs[v]).sum().backward()
        if v == rank:
            mine = out[0].detach()
    assert torch.equal(image, mine), "image differs from the single-GPU render"
    worst = {}
    for k in full:
        want = ref[k].grad[lo:hi]
        worst[k] = (loc[k].grad - want).abs().max().item() / max(want.abs().max().item(), 1e-30)
        assert worst[k] <= TOL_GRAD_REL, (k, worst[k])
    torch.cuda.synchronize()
    print(f"rank {rank}: OK", {k: f"{v:.1e}" for k, v in worst.items()}, flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
