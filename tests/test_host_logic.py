"""CPU-only tests of the host-side mirror of the reference interface (no kernels run)."""
import importlib
import os

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def dgr():
    import diff_gaussian_rasterization as m
    return m


def _settings(dgr, **kw):
    base = dict(image_height=32, image_width=48, tanfovx=0.5, tanfovy=0.4, bg=torch.zeros(3), scale_modifier=1.0,
                viewmatrix=torch.eye(4), projmatrix=torch.eye(4), sh_degree=3, campos=torch.zeros(3),
                prefiltered=False, debug=False)
    base.update(kw)
    return dgr.GaussianRasterizationSettings(**base)


def test_settings_accepts_reference_kwargs(dgr):
    s = _settings(dgr, antialiasing=True)     # the adapter passes it (gaussian_renderer/__init__.py:576)
    assert s.antialiasing is True and s.image_height == 32 and s.sh_degree == 3
    with pytest.raises(TypeError):            # unknown keyword -> TypeError, like any Python callable
        _settings(dgr, not_a_field=1)


def test_argument_validation(dgr):
    r = dgr.GaussianRasterizer(raster_settings=_settings(dgr))
    m, sh, op = torch.zeros(4, 3), torch.zeros(4, 16, 3), torch.zeros(4, 1)
    sc, q = torch.ones(4, 3), torch.zeros(4, 4)
    with pytest.raises(ValueError):
        r(means3D=m, means2D=None, shs=sh, colors_precomp=torch.zeros(4, 3), opacities=op, scales=sc, rotations=q)
    with pytest.raises(ValueError):
        r(means3D=m, means2D=None, shs=None, colors_precomp=None, opacities=op, scales=sc, rotations=q)
    with pytest.raises(ValueError):
        r(means3D=m, means2D=None, shs=sh, opacities=op, scales=sc, rotations=q, cov3D_precomp=torch.zeros(4, 6))
    with pytest.raises(ValueError):
        r(means3D=m, means2D=None, shs=sh, opacities=op)
    with pytest.raises(ValueError):   # degree 3 needs 16 coefficients
        r(means3D=m, means2D=None, shs=torch.zeros(4, 4, 3), opacities=op, scales=sc, rotations=q)
    # CPU tensors: refused loudly, never a silent fallback, never TypeError (the adapter would retry)
    with pytest.raises(RuntimeError, match="GPU"):
        r(means3D=m, means2D=None, shs=sh, opacities=op, scales=sc, rotations=q)


def test_synthetic_scene_is_deterministic():
    syn = importlib.import_module("gaussian-splatting_amd.synthetic")
    a, b = syn.make_scene(100, 16, 40, 60, seed=3), syn.make_scene(100, 16, 40, 60, seed=3)
    for k in ("means3D", "scales", "rotations", "opacities", "shs", "G"):
        assert torch.equal(getattr(a, k), getattr(b, k))
    assert float(a.opacities.min()) >= 0.01 and float(a.opacities.max()) <= 0.99
    assert torch.allclose(a.rotations.norm(dim=1), torch.ones(100), atol=1e-6)
    v = syn.jitter_view(a, 1)
    assert torch.equal(v.means3D, a.means3D) and not torch.equal(v.viewmatrix, a.viewmatrix)
    assert torch.allclose(v.viewmatrix[:3, 3], -v.campos)


def test_partition_helpers():
    par = importlib.import_module("gaussian-splatting_amd.parallel")
    for rows, world in ((135, 8), (53, 8), (7, 8), (13, 2), (1, 1)):
        bands = [par.tile_row_band(rows, r, world) for r in range(world)]
        assert bands[0][0] == 0 and bands[-1][1] == rows
        assert all(bands[i][1] == bands[i + 1][0] for i in range(world - 1))
        sizes = [e - b for b, e in bands]
        assert max(sizes) - min(sizes) <= 1
    assert par.tile_row_band(135, 0, 8) == (0, 17) and par.tile_row_band(135, 7, 8) == (119, 135)
    assert sorted(sum((par.shard_views(8, r, 3) for r in range(3)), [])) == list(range(8))


def test_byte_model_matches_survey():
    import bench
    b = bench.bytes_v1(3_000_000, 2_539_666, 7_535_647, 1297 * 840, 16, 82 * 53, 6)
    assert abs(sum(b.values()) / 1e9 - 4.99) < 0.02          # ~4.99 GB / frame
    assert round(b["radix_sort"] / 1e6) == 1145 and round(b["render_bwd"] / 1e6) == 788


def test_default_t_eps_from_environment():
    """GSPLAT_T_EPS replaces the default stop threshold at import (the reference's adapter cannot pass t_eps)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import importlib; r = importlib.import_module('gaussian-splatting_amd.rasterizer'); "
            "import inspect; print(r.DEFAULT_T_EPS, inspect.signature(r.GaussianRasterizer.__init__).parameters['t_eps'].default)")
    for env_val, want in (("0", "0.0 0.0"), ("1e-7", "1e-07 1e-07"), (None, "1e-05 1e-05")):
        env = dict(os.environ)
        env.pop("GSPLAT_T_EPS", None)
        if env_val is not None:
            env["GSPLAT_T_EPS"] = env_val
        out = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=120)
        assert out.returncode == 0, out.stderr
        assert out.stdout.strip() == want
    env = dict(os.environ, GSPLAT_T_EPS="2")
    out = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode != 0 and "GSPLAT_T_EPS" in out.stderr


def test_pair_capacity_cache_is_per_device_and_tile_grid_and_thread_safe():
    """SURVEY.md 8b "Threading": the one piece of mutable host state is a cache behind a lock; two image sizes (or
    two bands) on one device keep separate capacities, so alternating cameras do not trigger re-binning."""
    import threading
    rz = importlib.import_module("gaussian-splatting_amd.rasterizer")
    c = rz.PairCapacityCache()
    a, b = (0, 82, 53, 0, 53), (0, 50, 50, 0, 50)
    assert c.get(a) == 0
    assert c.learn(a, 7_535_647) >= 7_535_647
    assert c.learn(b, 755_918) >= 755_918
    assert c.get(a) >= 7_535_647 > c.get(b) >= 755_918          # the small frame did not shrink the large one's
    assert c.learn(a, 10) == c.get(a) >= 7_535_647              # never shrinks
    assert c.get((1,) + a[1:]) == 0 and c.get(a[:3] + (0, 7)) == 0   # other device / other band: nothing learnt

    def worker(k):
        for i in range(2000):
            c.learn((0, k, k, 0, k), i)

    ts = [threading.Thread(target=worker, args=(k,)) for k in range(1, 9)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert all(c.get((0, k, k, 0, k)) == 1999 + 1999 // 4 + 4096 for k in range(1, 9))
    assert isinstance(rz._pair_capacity, rz.PairCapacityCache)


def test_byte_model_of_the_factored_sort():
    """bench.py prices each sort launch on the bytes it moves: the v1 row charges 6 passes of 24 B over
    the D pairs, the build runs 4 passes over the N depth keys + 2 over the D tile ids."""
    import bench
    N, D = 3_000_000, 7_535_647
    assert bench.bytes_depth_sort(N) == 4 * 20 * N
    assert bench.bytes_tile_sort(D, 13) == 2 * 20 * D and bench.bytes_tile_sort(D, 8) == 20 * D
    b = bench.bytes_v1(N, 2_539_666, D, 1297 * 840, 16, 82 * 53, 6)
    own = sum(b.values()) - b["radix_sort"] + bench.bytes_tile_sort(D, 13) + bench.bytes_depth_sort(N)
    assert 4.3e9 < own < sum(b.values())


def test_view_sharded_step_argument_checks(dgr):
    par = importlib.import_module("gaussian-splatting_amd.parallel")
    with pytest.raises(ValueError):
        par.ViewShardedStep([])
    step = par.ViewShardedStep([_settings(dgr), _settings(dgr)])
    m, sh, op = torch.zeros(4, 3), torch.zeros(4, 16, 3), torch.zeros(4, 1)
    sc, q = torch.ones(4, 3), torch.zeros(4, 4)
    with pytest.raises(ValueError):       # both colour sources, as for GaussianRasterizer
        step(means3D=m, shs=sh, colors_precomp=torch.zeros(4, 3), opacities=op, scales=sc, rotations=q)
    with pytest.raises(RuntimeError, match="GPU"):
        step(means3D=m, shs=sh, opacities=op, scales=sc, rotations=q)
    assert par.shard_views(8, 1, 4) == [1, 5] and par.shard_views(8, 0, 1) == list(range(8))
    assert isinstance(par.ViewShardedRasterizer(_settings(dgr)), par.ViewShardedStep)
    with pytest.raises(ValueError, match="sh_degree"):     # one model, one active degree: the fused per-Gaussian launches take one
        par.ViewShardedStep([_settings(dgr), _settings(dgr, sh_degree=2)])


def test_multi_view_entry_points_host_checks(dgr):
    """rasterizer.preprocess_views / preprocess_backward_views (gs_preprocess_fwd_views / gs_preprocess_bwd_views): the
    argument checks that need no GPU, and the descriptor structs against the header's field order."""
    R = importlib.import_module("gaussian-splatting_amd.rasterizer")
    L = importlib.import_module("gaussian-splatting_amd._lib")
    m, sh, op = torch.zeros(4, 3), torch.zeros(4, 16, 3), torch.zeros(4, 1)
    sc, q = torch.ones(4, 3), torch.zeros(4, 4)
    two = [_settings(dgr), _settings(dgr, sh_degree=1)]
    with pytest.raises(ValueError, match="sh_degree"):
        R.preprocess_views(two, m, op, sc, q, None, sh, None)
    with pytest.raises(ValueError, match="sh_degree"):
        R.preprocess_backward_views(two, m, op, sc, q, None, sh, None, [None, None], [None, None], [True] * 8)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = open(os.path.join(root, "include", "gsplat.h")).read()
    for cls, name in ((L.GsFwdView, "GsFwdView"), (L.GsBwdView, "GsBwdView")):
        body = header[header.index("typedef struct " + name):header.index("} " + name + ";")]
        fields = []
        for line in body.splitlines()[1:]:
            line = line.strip().rstrip(";")
            if line:
                fields += [f.strip(" *") for f in line.split(" ", 1)[1].replace("float*", "").replace("void*", "").split(",")]
        fields = [f.split("*")[-1].strip() for f in fields]
        assert [f[0] for f in cls._fields_] == fields, (name, fields)
    assert L.MAX_BWD_VIEWS == 8 and "#define GS_MAX_BWD_VIEWS 8" in header and "#define GS_MAX_FWD_VIEWS 8" in header
