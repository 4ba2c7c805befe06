"""CPU, world_size 2, gloo: the multi-GPU logic of the path.

 * view-sharded training step: each rank differentiates its own camera view of the same replicated
   Gaussians (here with the CPU oracle standing in for the per-rank HIP call), the bucketed
   all-reduce sums the per-Gaussian gradients; result == sequential accumulation of the same views.
 * densification statistics: SUM / SUM / MAX.
 * tile-row sharded frame: bands gathered on rank 0 == the whole frame.
 * Gaussian-sharded step (SplatExchange): shard bookkeeping and the uneven all-to-all both ways (the
   kernels around it need a GPU: tests/test_gpu_exchange.py).
"""
import importlib
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _view_grads(sc, G):
    from oracle import oracle
    a = lambda t: t.numpy()  # noqa: E731
    outs, grads, geom = oracle.forward_backward(
        sc.H, sc.W, sc.sh_degree, a(sc.viewmatrix), a(sc.projmatrix), a(sc.campos), a(sc.bg), a(sc.means3D),
        a(sc.opacities), scales=a(sc.scales), rotations=a(sc.rotations), shs=a(sc.shs), G_C=a(G))
    return outs, {k: torch.from_numpy(grads[k]) for k in ("means3D", "shs", "opacities", "scales", "rotations")}, geom


def _worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        syn = importlib.import_module("gaussian-splatting_amd.synthetic")
        par = importlib.import_module("gaussian-splatting_amd.parallel")
        base = syn.make_scene(400, 16, 48, 64, seed=2)
        names = ("means3D", "shs", "opacities", "scales", "rotations")
        # ---- view-sharded gradients, 4 views over 2 ranks, tiny buckets to force several per tensor
        params = {k: getattr(base, k).clone().requires_grad_(True) for k in names}
        mine = par.shard_views(4, rank, world)
        for k in names:
            params[k].grad = torch.zeros_like(params[k])
        for v in mine:
            _, g, _ = _view_grads(syn.jitter_view(base, v), base.G)
            for k in names:
                params[k].grad += g[k]
        par.GradReducer(list(params.values()), bucket_bytes=4096).all_reduce_grads()
        ref = {k: torch.zeros_like(params[k]) for k in names}
        for v in range(4):
            _, g, _ = _view_grads(syn.jitter_view(base, v), base.G)
            for k in names:
                ref[k] += g[k]
        for k in names:
            err = (params[k].grad - ref[k]).abs().max().item()
            assert err <= 1e-5 * ref[k].abs().max().item() + 1e-12, (k, err)
        # ---- densification statistics
        acc = torch.full((10, 1), float(rank + 1))
        den = torch.full((10, 1), 1.0)
        rad = torch.arange(10, dtype=torch.float32) * (rank + 1)
        par.reduce_densify_stats(acc, den, rad)
        assert torch.all(acc == 3.0) and torch.all(den == 2.0) and torch.equal(rad, torch.arange(10.0) * 2)
        # ---- tile-row bands: H = 48 -> 3 tile rows -> ranks own 2 and 1
        frame = torch.arange(3 * 48 * 64, dtype=torch.float32).reshape(3, 48, 64)
        b, e = par.tile_row_band(3, rank, world)
        full = par.gather_bands(frame[:, b * 16:e * 16].contiguous(), 48, dst=0)
        if rank == 0:
            assert torch.equal(full, frame)
        else:
            assert full is None
        # ---- splat exchange plumbing: shard sizes, offsets, uneven all-to-all in global index order
        counts = [3, 5]
        ex = par.SplatExchange(counts[rank])
        assert ex.counts == counts and ex.offset == sum(counts[:rank])
        n = counts[rank]
        # row (v, i) of my send buffer = "Gaussian offset+i seen from view v"
        send = torch.stack([torch.arange(ex.offset, ex.offset + n, dtype=torch.float32) + 100.0 * v
                            for v in range(world)]).reshape(world * n, 1)
        recv = torch.empty((sum(counts), 1))
        par._all_to_all_rows(recv, send, counts, [n] * world, None)
        assert torch.equal(recv[:, 0], torch.arange(8, dtype=torch.float32) + 100.0 * rank)
        back = torch.empty((world * n, 1))
        par._all_to_all_rows(back, recv, [n] * world, counts, None)     # the backward's direction
        assert torch.equal(back, send)
        open(os.path.join(tmp, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_world_size_2_gloo(tmp_path, oracle):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok0").exists() and (tmp_path / "ok1").exists()
