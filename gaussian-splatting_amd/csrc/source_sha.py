#!/usr/bin/env python3
"""Hash of the kernel sources libgsplat_hip.so is built from (csrc/*.hip, csrc/*.h, include/gsplat.h).  The Makefile compiles it
into the library (gs_source_sha()); __graft_entry__.build() and bench.py compare it with the tree they run from, and
profiles/counters.json is stamped with it."""
import glob
import hashlib
import os


def source_sha() -> str:
    d = os.path.dirname(os.path.abspath(__file__))
    files = sorted(glob.glob(os.path.join(d, "*.hip")) + glob.glob(os.path.join(d, "*.h")))
    files.append(os.path.join(d, "..", "..", "include", "gsplat.h"))
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(source_sha())
