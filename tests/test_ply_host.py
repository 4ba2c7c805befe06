"""CPU: the host half of the PLY row -- header parser and header writer of the C-ABI against
the numpy oracle (oracle/ply_oracle.py).  The reference reads and writes through `plyfile`, which is not installed
here and no .ply fixture ships with the reference: "parity unpinned" at the reference, pinned library <-> oracle."""
import importlib
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ply_oracle  # noqa: E402


def make_model(N, K, seed=0):
    g = np.random.default_rng(seed)
    f = lambda *s: g.standard_normal(s).astype(np.float32)  # noqa: E731
    return dict(xyz=f(N, 3), features_dc=f(N, 1, 3), features_rest=f(N, K - 1, 3), opacity=f(N, 1), scaling=f(N, 3),
                rotation=f(N, 4))


@pytest.fixture(scope="module")
def ply_io():
    return importlib.import_module("gaussian-splatting_amd.ply_io")


@pytest.mark.parametrize("K", [1, 4, 9, 16])
def test_oracle_round_trip_and_header_layout(ply_io, K):
    N = 37
    m = make_model(N, K)
    data = ply_oracle.save_ply_bytes(**m)
    back = ply_oracle.load_ply_bytes(data)
    for k in m:
        assert np.array_equal(back[k], m[k]), k
    L = ply_io.parse_header(data)
    names = ply_oracle.attribute_names(3 * (K - 1))
    assert L.num_vertices == N and L.stride_bytes == 4 * len(names) and L.sh_rest_count == 3 * (K - 1)
    assert L.data_offset == data.index(b"end_header\n") + 11
    off = {n: 4 * i for i, n in enumerate(names)}
    assert list(L.off_xyz) == [off["x"], off["y"], off["z"]]
    assert list(L.off_f_dc) == [off[f"f_dc_{i}"] for i in range(3)]
    assert L.off_opacity == off["opacity"]
    assert list(L.off_scale) == [off[f"scale_{i}"] for i in range(3)]
    assert list(L.off_rot) == [off[f"rot_{i}"] for i in range(4)]
    assert list(L.off_f_rest)[:3 * (K - 1)] == [off[f"f_rest_{i}"] for i in range(3 * (K - 1))]


def test_header_writer_equals_oracle(ply_io):
    import ctypes as C
    lib = importlib.import_module("gaussian-splatting_amd._lib").load()
    for N, R in ((0, 0), (5, 45), (123456789, 24)):
        n = lib.gs_ply_write_header(None, 0, N, R, None)
        buf = C.create_string_buffer(int(n))
        assert lib.gs_ply_write_header(buf, n, N, R, None) == n
        want = "ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % N + \
            "".join("property float %s\n" % a for a in ply_oracle.attribute_names(R)) + "end_header\n"
        assert buf.raw[:n] == want.encode()


def test_parser_other_orders_extra_properties_and_errors(ply_io):
    head = (b"ply\nformat binary_little_endian 1.0\ncomment made by hand\nelement vertex 2\n"
            b"property uchar red\nproperty uchar green\nproperty uchar blue\nproperty uchar pad\n"
            b"property float rot_0\nproperty float rot_1\nproperty float rot_2\nproperty float rot_3\n"
            b"property double weight\nproperty float opacity\n"
            b"property float z\nproperty float y\nproperty float x\n"
            b"property float f_dc_2\nproperty float f_dc_1\nproperty float f_dc_0\n"
            b"property float scale_0\nproperty float scale_1\nproperty float scale_2\n"
            b"element face 0\nproperty list uchar int vertex_indices\nend_header\n")
    stride = 4 + 16 + 8 + 4 + 12 + 12 + 12
    L = ply_io.parse_header(head + b"\0" * (2 * stride))
    assert L.stride_bytes == stride and L.sh_rest_count == 0 and list(L.off_rot) == [4, 8, 12, 16]
    assert L.off_opacity == 28 and list(L.off_xyz) == [40, 36, 32] and list(L.off_f_dc) == [52, 48, 44]
    with pytest.raises(RuntimeError, match="too short"):
        ply_io.parse_header(head + b"\0" * (2 * stride - 1))
    with pytest.raises(RuntimeError, match="binary_little_endian"):
        ply_io.parse_header(head.replace(b"binary_little_endian", b"ascii"))
    with pytest.raises(RuntimeError, match="magic"):
        ply_io.parse_header(b"plx\n" + head[4:])
    with pytest.raises(RuntimeError, match="end_header"):
        ply_io.parse_header(head[:-11])
    with pytest.raises(RuntimeError, match="lacks"):
        ply_io.parse_header(head.replace(b"property float opacity\n", b"property float opacitx\n") + b"\0" * 200)
