"""CPU oracle of the trained-model PLY I/O.  TEST INFRASTRUCTURE: only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

numpy restatement of GaussianModel.save_ply / load_ply (reference scene/gaussian_model.py:333-375, :377-459;
attribute order :316-331).  The reference goes through the third-party `plyfile` package (not installed in this
image, and no .ply fixture ships with the reference), so this row is "parity unpinned" at the reference: the
file format is restated from the PLY specification as plyfile 1.0 writes it for a single all-float32 "vertex"
element -- "ply", "format binary_little_endian 1.0", "element vertex N", one "property float <name>" per
attribute, "end_header", then the records.
"""
import numpy as np


def attribute_names(n_rest):
    """reference :316-331"""
    names = ["x", "y", "z", "nx", "ny", "nz"] + [f"f_dc_{i}" for i in range(3)] + [f"f_rest_{i}" for i in range(n_rest)]
    return names + ["opacity"] + [f"scale_{i}" for i in range(3)] + [f"rot_{i}" for i in range(4)]


def save_ply_bytes(xyz, features_dc, features_rest, opacity, scaling, rotation):
    """reference :333-375.  features_dc (N,1,3), features_rest (N,K-1,3): written channel-major."""
    N = xyz.shape[0]
    f_dc = np.ascontiguousarray(np.transpose(features_dc, (0, 2, 1))).reshape(N, features_dc.shape[1] * 3)
    f_rest = np.ascontiguousarray(np.transpose(features_rest, (0, 2, 1))).reshape(N, features_rest.shape[1] * 3)
    attrs = np.concatenate((xyz, np.zeros_like(xyz), f_dc, f_rest, opacity.reshape(N, 1), scaling, rotation), axis=1)
    names = attribute_names(f_rest.shape[1])
    assert attrs.shape[1] == len(names)
    header = "ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % N
    header += "".join("property float %s\n" % n for n in names) + "end_header\n"
    return header.encode("ascii") + np.ascontiguousarray(attrs, dtype="<f4").tobytes()


_TYPES = {"char": "i1", "uchar": "u1", "int8": "i1", "uint8": "u1", "short": "<i2", "ushort": "<u2", "int16": "<i2",
          "uint16": "<u2", "int": "<i4", "uint": "<u4", "int32": "<i4", "uint32": "<u4", "float": "<f4", "float32": "<f4",
          "double": "<f8", "float64": "<f8"}


def parse(data: bytes):
    """-> (structured array of the vertex element, property names in file order)"""
    end = data.index(b"end_header\n") + len(b"end_header\n")
    lines = data[:end].decode("ascii").split("\n")
    assert lines[0] == "ply"
    fields, n, in_vertex = [], None, False
    for ln in lines[1:]:
        t = ln.split()
        if not t or t[0] in ("comment", "obj_info"):
            continue
        if t[0] == "format":
            assert t[1] == "binary_little_endian"
        elif t[0] == "element":
            in_vertex = t[1] == "vertex"
            if in_vertex:
                n = int(t[2])
        elif t[0] == "property" and in_vertex:
            fields.append((t[2], _TYPES[t[1]]))
    arr = np.frombuffer(data, dtype=np.dtype(fields), count=n, offset=end)
    return arr, [f[0] for f in fields]


def load_ply_bytes(data: bytes):
    """reference :377-459 -> dict of float32 arrays shaped as the reference's parameters."""
    v, names = parse(data)
    N = v.shape[0]
    xyz = np.stack((v["x"], v["y"], v["z"]), axis=1).astype(np.float32)
    opacity = np.asarray(v["opacity"], dtype=np.float32)[:, None]
    f_dc = np.zeros((N, 3, 1), np.float32)
    for c in range(3):
        f_dc[:, c, 0] = v[f"f_dc_{c}"]
    rest_names = [n for n in names if n.startswith("f_rest_")]          # file order, as the reference
    assert len(rest_names) % 3 == 0
    f_rest = np.zeros((N, len(rest_names)), np.float32)
    for i, n in enumerate(rest_names):
        f_rest[:, i] = v[n]
    f_rest = f_rest.reshape(N, 3, len(rest_names) // 3)
    scale_names = [n for n in names if n.startswith("scale_")]
    rot_names = [n for n in names if n.startswith("rot")]
    scaling = np.stack([v[n] for n in scale_names], axis=1).astype(np.float32)
    rotation = np.stack([v[n] for n in rot_names], axis=1).astype(np.float32)
    return dict(xyz=xyz, features_dc=np.ascontiguousarray(f_dc.transpose(0, 2, 1)),
                features_rest=np.ascontiguousarray(f_rest.transpose(0, 2, 1)), opacity=opacity, scaling=scaling,
                rotation=rotation)
