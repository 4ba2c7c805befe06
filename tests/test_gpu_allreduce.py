"""GPU: the view-sharded training step BASELINE.json's north_star names (replicated Gaussians, one view per rank,
bucketed gradient all-reduce) with two / three ranks on the box's one GPU and the HIP kernels doing the rendering.
tests/allreduce_worker.py holds the checks: gradients == sequential accumulation of the same views on one process
(<= 1e-4 of max|grad|), for the all-reduce overlapped with the backward (parallel.ViewShardedRasterizer) and for
the post-hoc parallel.GradReducer; statistics through reduce_densify_stats."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.gpu
@pytest.mark.parametrize("world,variant", [(2, "sh"), (3, "sh"), (2, "precomp")])
def test_view_sharded_allreduce_equals_sequential_accumulation(world, variant):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   ALLREDUCE_VARIANT=variant)
        procs.append(subprocess.Popen([sys.executable, "-u", os.path.join(ROOT, "tests", "allreduce_worker.py")],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=100)[0])
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} failed:\n" + "\n".join(f"--- rank {k} ---\n{x[-2500:]}" for k, x in enumerate(outs))
        assert f"rank {r}: OK" in o
