"""gs_sort_pairs_static (the entry point that allocates nothing: its histogram tables live in a
file-scope device array) against gs_sort_pairs on the same inputs, through the C-ABI, bit for bit.
Two calls in a row share the device's scratch array; the second is smaller, so stale table rows of
the first call lie behind the ones it uses.  Collected last on purpose (see the file name)."""
import ctypes as C
import importlib

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _sort(L, entry, keys, vals, n, cap, bits, use_count, dev):
    t = lambda a: torch.from_numpy(a.view(np.int32).copy()).to(dev)  # noqa: E731
    ka, va = t(keys), t(vals)
    kb, vb = torch.empty_like(ka), torch.empty_like(va)
    cnt = torch.tensor([n], dtype=torch.int32, device=dev) if use_count else None
    in_b = C.c_int(0)
    p = lambda x: None if x is None else C.c_void_p(x.data_ptr())  # noqa: E731
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    if entry == "static":
        torch.cuda.synchronize()
        before = torch.cuda.memory_stats(dev)["allocation.all.allocated"], torch.cuda.mem_get_info(dev)[0]
        rc = L.gs_sort_pairs_static(cap, p(cnt), 0, bits, p(ka), p(va), p(kb), p(vb), C.byref(in_b), st)
        torch.cuda.synchronize()
        after = torch.cuda.memory_stats(dev)["allocation.all.allocated"], torch.cuda.mem_get_info(dev)[0]
        assert after[0] == before[0] and after[1] >= before[1], "the entry point allocated device memory"
    else:
        ws_bytes = L.gs_sort_workspace_bytes(cap)
        ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=dev)
        rc = L.gs_sort_pairs(cap, p(cnt), 0, bits, p(ka), p(va), p(kb), p(vb), p(ws), ws_bytes, C.byref(in_b), st)
        torch.cuda.synchronize()
    assert rc == 0, L.gs_last_error()
    ko, vo = (kb, vb) if in_b.value else (ka, va)
    return ko.cpu().numpy().view(np.uint32)[:n].copy(), vo.cpu().numpy().view(np.uint32)[:n].copy()


def test_static_sort_equals_workspace_sort_two_calls_in_a_row():
    L = importlib.import_module("gaussian-splatting_amd._lib").load()
    dev = torch.device("cuda:0")
    # (pairs, capacity slack, key bits, count on the device): large first, then smaller ones
    cases = [(1_500_003, 0, 32, False), (70_001, 5000, 13, True), (4097, 0, 32, False), (1, 0, 8, False)]
    with torch.cuda.device(dev):
        torch.zeros(1, device=dev)                       # context
        # the documented set-up call (include/gsplat.h): D = 0, first call on the device -- when this file runs alone it is
        # the library's first HIP call altogether (code object load, symbol address); every later call must allocate nothing
        in_b = C.c_int(-1)
        assert L.gs_sort_pairs_static(0, None, 0, 32, None, None, None, None, C.byref(in_b), None) == 0 and in_b.value == 0
        for n, slack, bits, use_count in cases:
            rng = np.random.default_rng(n)
            cap = n + slack
            keys = rng.integers(0, 2 ** 32, size=cap, dtype=np.uint64).astype(np.uint32)
            keys[: n // 2] &= np.uint32(0x3FF)           # many equal keys: stability matters
            vals = np.arange(cap, dtype=np.uint32)
            k_ref, v_ref = _sort(L, "workspace", keys, vals, n, cap, bits, use_count, dev)
            k_st, v_st = _sort(L, "static", keys, vals, n, cap, bits, use_count, dev)
            assert np.array_equal(k_st, k_ref) and np.array_equal(v_st, v_ref), (n, bits)
            mask = np.uint32((1 << bits) - 1) if bits < 32 else np.uint32(0xFFFFFFFF)
            perm = np.argsort(keys[:n] & mask, kind="stable")
            assert np.array_equal(k_st, keys[:n][perm]) and np.array_equal(v_st, vals[:n][perm])


def test_static_sort_rejects_sizes_beyond_its_scratch():
    L = importlib.import_module("gaussian-splatting_amd._lib").load()
    limit = L.gs_sort_pairs_static_max()
    assert limit == 1 << 24
    in_b = C.c_int(0)
    d = C.c_void_p(256)     # never dereferenced: the size check comes first
    rc = L.gs_sort_pairs_static(limit + 1, None, 0, 32, d, d, d, d, C.byref(in_b), None)
    assert rc == -3 and b"exceed" in L.gs_last_error()
