"""GPU: PLY unpack / pack kernels (gs_ply_unpack, gs_ply_pack through ply_io) against the numpy oracle, bit for bit;
the activated outputs against torch's cat / sigmoid / exp / normalize (tolerance 2e-6 relative: libm-vs-torch exp)."""
import importlib
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ply_oracle  # noqa: E402
from test_ply_host import make_model  # noqa: E402

pytestmark = pytest.mark.gpu
TOL_ACT = 2e-6


@pytest.fixture(scope="module")
def ply_io():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return importlib.import_module("gaussian-splatting_amd.ply_io")


@pytest.mark.parametrize("N,K", [(0, 16), (1, 16), (63, 16), (64, 1), (1000, 4), (70001, 16), (257, 9)])
def test_load_equals_oracle_and_save_round_trip(ply_io, tmp_path, N, K):
    m = make_model(N, K, seed=N + K)
    data = ply_oracle.save_ply_bytes(**m)
    path = str(tmp_path / "point_cloud.ply")
    open(path, "wb").write(data)
    got = ply_io.load_ply(path, "cuda:0", activated=True)
    want = ply_oracle.load_ply_bytes(data)
    for k in want:
        assert tuple(got[k].shape) == want[k].shape, k
        assert np.array_equal(got[k].cpu().numpy(), want[k]), k
    assert got["sh_degree"] == int(round(K ** 0.5)) - 1
    # activated tensors = what GaussianModel's properties hand to the rasterizer (reference :104-142)
    t = {k: torch.from_numpy(v.copy()) for k, v in want.items()}
    shs = torch.cat((t["features_dc"], t["features_rest"]), dim=1)
    assert torch.equal(got["shs"].cpu(), shs)
    for key, ref in (("opacities", torch.sigmoid(t["opacity"])), ("scales", torch.exp(t["scaling"])),
                     ("rotations", torch.nn.functional.normalize(t["rotation"]))):
        err = ((got[key].cpu() - ref).abs() / ref.abs().clamp_min(1e-30)).max().item() if N else 0.0
        assert err <= TOL_ACT, (key, err)
    # save: the bytes the oracle writes
    out = str(tmp_path / "sub" / "saved.ply")
    ply_io.save_ply(out, *[got[k] for k in ply_io.RAW])
    assert open(out, "rb").read() == data


def test_shuffled_properties_and_extra_columns(ply_io, tmp_path):
    """Offsets come from the header, not from the reference's attribute order."""
    N, K = 300, 4
    m = make_model(N, K, seed=5)
    base, names = ply_oracle.parse(ply_oracle.save_ply_bytes(**m))
    rng = np.random.default_rng(1)
    order = list(names)
    rest = [n for n in order if n.startswith("f_rest_")]
    others = [n for n in order if not n.startswith("f_rest_")]
    rng.shuffle(others)
    new_names = others[:5] + rest + others[5:]                  # f_rest keeps its file order (load_ply keys on it)
    fields = [("tag", "<u4")] + [(n, "<f4") for n in new_names] + [("w", "<f8")]
    arr = np.zeros(N, dtype=np.dtype(fields))
    for n in names:
        arr[n] = base[n]
    arr["tag"] = 0xDEADBEEF
    header = "ply\nformat binary_little_endian 1.0\nelement vertex %d\nproperty uint tag\n" % N + \
        "".join("property float %s\n" % n for n in new_names) + "property double w\nend_header\n"
    data = header.encode() + arr.tobytes()
    path = str(tmp_path / "shuffled.ply")
    open(path, "wb").write(data)
    got = ply_io.load_ply(path, "cuda:0")
    want = ply_oracle.load_ply_bytes(data)
    for k in want:
        assert np.array_equal(got[k].cpu().numpy(), want[k]), k
        if k not in ("scaling", "rotation"):      # scale_* / rot_* are taken in FILE order (reference :410-424),
            assert np.array_equal(want[k], m[k]), k   # so shuffling them permutes the columns, here as there


@pytest.mark.parametrize("stride", [256, 512])
def test_power_of_two_record_strides(ply_io, tmp_path, stride):
    """Records of exactly 256 / 512 bytes (a degree-3 model plus extra float columns): the staging area of a
    workgroup is sized to leave room for the kernel's static offset table inside the 64 KB of LDS."""
    N, K = 1000, 16
    m = make_model(N, K, seed=6)
    base, names = ply_oracle.parse(ply_oracle.save_ply_bytes(**m))
    extra = [f"extra_{i}" for i in range((stride - 4 * len(names)) // 4)]
    fields = [(n, "<f4") for n in names + extra]
    arr = np.zeros(N, dtype=np.dtype(fields))
    assert arr.dtype.itemsize == stride
    for n in names:
        arr[n] = base[n]
    header = "ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % N + \
        "".join("property float %s\n" % n for n in names + extra) + "end_header\n"
    data = header.encode() + arr.tobytes()
    path = str(tmp_path / f"stride{stride}.ply")
    open(path, "wb").write(data)
    got = ply_io.load_ply(path, "cuda:0")
    want = ply_oracle.load_ply_bytes(data)
    for k in want:
        assert np.array_equal(got[k].cpu().numpy(), want[k]), k


def test_loaded_model_renders(ply_io, tmp_path):
    """End of the row: file -> device tensors -> the rasterizer, no host-side reshuffling in between."""
    impl = importlib.import_module("gaussian-splatting_amd")
    syn = importlib.import_module("gaussian-splatting_amd.synthetic")
    dev = torch.device("cuda:0")
    sc = syn.make_scene(2000, 16, 96, 128, seed=9)
    m = dict(xyz=sc.means3D.numpy(), features_dc=sc.shs[:, :1].numpy(), features_rest=sc.shs[:, 1:].numpy(),
             opacity=torch.logit(sc.opacities).numpy(), scaling=torch.log(sc.scales).numpy(), rotation=(sc.rotations * 3).numpy())
    path = str(tmp_path / "m.ply")
    open(path, "wb").write(ply_oracle.save_ply_bytes(**m))
    g = ply_io.load_ply(path, dev, activated=True)
    st = impl.GaussianRasterizationSettings(
        image_height=sc.H, image_width=sc.W, tanfovx=sc.tanfovx, tanfovy=sc.tanfovy, bg=sc.bg.to(dev), scale_modifier=1.0,
        viewmatrix=sc.viewmatrix.to(dev), projmatrix=sc.projmatrix.to(dev), sh_degree=sc.sh_degree, campos=sc.campos.to(dev),
        prefiltered=False, debug=False)
    rast = impl.GaussianRasterizer(st)
    a = rast(means3D=g["xyz"], shs=g["shs"], opacities=g["opacities"], scales=g["scales"], rotations=g["rotations"])[0]
    b = rast(means3D=sc.means3D.to(dev), shs=sc.shs.to(dev), opacities=sc.opacities.to(dev), scales=sc.scales.to(dev),
             rotations=sc.rotations.to(dev))[0]
    assert (a - b).abs().max().item() <= 1e-4
