"""GPU: the Gaussian-sharded training step (parallel.SplatExchange) with two ranks on the box's one GPU.

Each rank owns a shard of the Gaussians and renders one view; splat records go to the renderers and
gradient rows come back.  tests/exchange_worker.py holds the checks (image
bit-identical to the single-GPU render, gradients within summation-order tolerance)."""
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
@pytest.mark.parametrize("world,variant", [(2, "sh"), (3, "sh"), (2, "precomp"), (2, "empty_shard")])
def test_splat_exchange_equals_single_gpu(world, variant):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   EXCHANGE_VARIANT=variant)
        procs.append(subprocess.Popen([sys.executable, "-u", os.path.join(ROOT, "tests", "exchange_worker.py")],
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
