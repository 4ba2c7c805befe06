"""Worker of tests/test_gpu_allreduce.py (not collected by pytest): one rank of the view-sharded training step.

Launched `world` times with RANK / WORLD_SIZE / MASTER_PORT in the environment; every rank uses cuda:0 (the GPU
box has one GPU) and the gloo backend (RCCL refuses two ranks on one device), so the collectives are staged
through the host -- the step's logic and kernels are the ones the RCCL run uses.  Every rank holds all
Gaussians and renders its own view with the HIP rasterizer:
  * ViewShardedRasterizer (bucketed all-reduce launched per Gaussian range inside the backward) and
  * the plain rasterizer followed by GradReducer.all_reduce_grads()
must both give the gradients of sequential accumulation of all views on one process (fp32 summation order:
<= 1e-4 of max|grad| per tensor); the image is the single-GPU image of the rank's view bit for bit; means2D's
gradient stays per view; reduce_densify_stats sums / maxes the statistics over the ranks.
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
    variant = os.environ.get("ALLREDUCE_VARIANT", "sh")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    impl = importlib.import_module("gaussian-splatting_amd")
    syn = importlib.import_module("gaussian-splatting_amd.synthetic")
    par = importlib.import_module("gaussian-splatting_amd.parallel")
    N, K, H, W = 30011, 16, 200, 312          # N is not a multiple of the bucket alignment
    base = syn.make_scene(N, K, H, W, seed=5)
    views = [syn.jitter_view(base, v) for v in range(world)]

    def settings(v):
        return impl.GaussianRasterizationSettings(
            image_height=H, image_width=W, tanfovx=v.tanfovx, tanfovy=v.tanfovy, bg=v.bg.to(dev), scale_modifier=1.0,
            viewmatrix=v.viewmatrix.to(dev), projmatrix=v.projmatrix.to(dev), sh_degree=v.sh_degree,
            campos=v.campos.to(dev), prefiltered=False, debug=False)

    sets = [settings(v) for v in views]
    gen = torch.Generator().manual_seed(200)
    Gs = [(torch.rand(3, H, W, generator=gen) * 2 - 1).to(dev) for _ in range(world)]
    Gd = [(torch.rand(1, H, W, generator=gen) * 0.1).to(dev) for _ in range(world)]
    if variant == "precomp":
        g = torch.Generator().manual_seed(78)
        Rm = torch.randn(N, 3, 3, generator=g) * base.scales.mean().item()
        S = Rm @ Rm.transpose(1, 2) + 1e-6 * torch.eye(3)
        data = {"means3D": base.means3D, "colors_precomp": torch.rand(N, 3, generator=g), "opacities": base.opacities,
                "cov3D_precomp": torch.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], 1).contiguous()}
    else:
        data = {k: getattr(base, k) for k in ("means3D", "shs", "opacities", "scales", "rotations")}

    def leaves():
        return {k: v.to(dev).clone().requires_grad_(True) for k, v in data.items()}

    def loss_of(out, v):
        return (out[0] * Gs[v]).sum() + (out[2] * Gd[v]).sum()

    # ---- single-process reference: every view, gradients accumulated in view order
    ref = leaves()
    ref_m2d, ref_img, ref_radii = [], None, []
    for v in range(world):
        mv = torch.zeros((N, 3), device=dev, requires_grad=True)
        out = impl.GaussianRasterizer(sets[v])(means2D=mv, **ref)
        loss_of(out, v).backward()
        ref_m2d.append(mv.grad)
        ref_radii.append(out[1])
        if v == rank:
            ref_img = out[0].detach()

    def check(tag, got, m2d):
        worst = {}
        for k in data:
            want = ref[k].grad
            worst[k] = (got[k].grad - want).abs().max().item() / max(want.abs().max().item(), 1e-30)
            assert worst[k] <= TOL_GRAD_REL, (tag, k, worst[k])
        assert torch.equal(m2d.grad, ref_m2d[rank]), tag + ": means2D's gradient must stay this view's own"
        return worst

    # ---- (1) bucketed all-reduce inside the backward
    mine = leaves()
    m2d = torch.zeros((N, 3), device=dev, requires_grad=True)
    rast = par.ViewShardedRasterizer(sets[rank], num_buckets=5)
    out = rast(means2D=m2d, **mine)
    assert torch.equal(out[0], ref_img), "image differs from the single-GPU render"
    loss_of(out, rank).backward()
    w1 = check("overlapped", mine, m2d)
    # ---- (2) plain rasterizer, then the post-hoc reducer (small buckets: several per tensor)
    mine2 = leaves()
    m2d2 = torch.zeros((N, 3), device=dev, requires_grad=True)
    out = impl.GaussianRasterizer(sets[rank])(means2D=m2d2, **mine2)
    loss_of(out, rank).backward()
    par.GradReducer(list(mine2.values()), bucket_bytes=1 << 20).all_reduce_grads()
    w2 = check("reducer", mine2, m2d2)
    # ---- densification statistics (train.py:185-193, scene/gaussian_model.py:701-713) over the ranks
    accum = m2d.grad[:, :2].norm(dim=-1, keepdim=True) * (out[1] > 0)[:, None]
    denom = (out[1] > 0).float()[:, None]
    maxr = out[1].float()
    par.reduce_densify_stats(accum, denom, maxr)
    want_acc = sum(ref_m2d[v][:, :2].norm(dim=-1, keepdim=True) * (ref_radii[v] > 0)[:, None] for v in range(world))
    assert torch.allclose(accum, want_acc, rtol=1e-6, atol=0)
    assert torch.equal(denom, sum((ref_radii[v] > 0).float()[:, None] for v in range(world)))
    assert torch.equal(maxr, torch.stack([r.float() for r in ref_radii]).max(dim=0).values)
    # ---- (3) BASELINE.json config 4's pattern: several views per rank and step, ONE exchange per step.  2 * world views,
    # rank r renders views r and r + world (parallel.shard_views); local views add up in place, the last one's backward
    # carries the bucketed all-reduce.  Reference: one process accumulating all 2 * world views in view order.
    nv = 2 * world
    views2 = [syn.jitter_view(base, 10 + v) for v in range(nv)]
    sets2 = [settings(v) for v in views2]
    Gs2 = [(torch.rand(3, H, W, generator=gen) * 2 - 1).to(dev) for _ in range(nv)]
    ref2 = leaves()
    ref2_m2d, ref2_img = {}, {}
    for v in range(nv):
        mv = torch.zeros((N, 3), device=dev, requires_grad=True)
        out = impl.GaussianRasterizer(sets2[v])(means2D=mv, **ref2)
        (out[0] * Gs2[v]).sum().backward()
        ref2_m2d[v], ref2_img[v] = mv.grad, out[0].detach()
    my_views = par.shard_views(nv, rank, world)
    assert len(my_views) == 2
    mine3 = leaves()
    m2d3 = torch.zeros((len(my_views), N, 3), device=dev, requires_grad=True)
    step = par.ViewShardedStep([sets2[v] for v in my_views], num_buckets=3)
    images, radii3, _, _ = step(means2D=m2d3, **mine3)
    for j, v in enumerate(my_views):
        assert torch.equal(images[j], ref2_img[v]), "multi-view step: image differs from the single-GPU render"
    sum((images[j] * Gs2[v]).sum() for j, v in enumerate(my_views)).backward()
    w3 = {}
    for k in data:
        want = ref2[k].grad
        w3[k] = (mine3[k].grad - want).abs().max().item() / max(want.abs().max().item(), 1e-30)
        assert w3[k] <= TOL_GRAD_REL, ("multi-view step", k, w3[k])
    for j, v in enumerate(my_views):
        assert torch.equal(m2d3.grad[j], ref2_m2d[v]), "multi-view step: means2D's gradient must stay per view"
    # only the first view's output used: the other view contributes nothing, the collective sequence is unchanged
    mine4 = leaves()
    images, _, _, _ = par.ViewShardedStep([sets2[v] for v in my_views], num_buckets=3)(**mine4)
    (images[0] * Gs2[my_views[0]]).sum().backward()
    ref4 = leaves()
    for v in range(world):        # the first local view of every rank: views 0 .. world - 1
        (impl.GaussianRasterizer(sets2[v])(**ref4)[0] * Gs2[v]).sum().backward()
    for k in data:
        e = (mine4[k].grad - ref4[k].grad).abs().max().item() / max(ref4[k].grad.abs().max().item(), 1e-30)
        assert e <= TOL_GRAD_REL, ("multi-view step, one output unused", k, e)
    torch.cuda.synchronize()
    print(f"rank {rank}: multi-view step", {k: f"{v:.1e}" for k, v in w3.items()}, flush=True)
    print(f"rank {rank}: OK overlapped", {k: f"{v:.1e}" for k, v in w1.items()}, "reducer",
          {k: f"{v:.1e}" for k, v in w2.items()}, flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
