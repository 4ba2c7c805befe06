"""Per-stage timing with HIP events on the launch stream (torch.cuda.Event on torch's current stream,
which is the stream every C-ABI call is given).  Off by default; bench.py turns it on."""
from __future__ import annotations

from collections import defaultdict

import torch

_enabled = False
_only = None
_events = defaultdict(list)


def enable(only=None):
    """only: iterable of stage names to time (None = all).  Two event records cost ~10 us of stream
    time per stage, so a throughput measurement times only the stage it reports on."""
    global _enabled, _only
    _events.clear()
    _only = None if only is None else frozenset(only)
    _enabled = True


def disable_and_collect(with_counts: bool = False):
    """Mean milliseconds per call of every stage recorded since enable() (and, on request, the call counts)."""
    global _enabled
    _enabled = False
    torch.cuda.synchronize()
    out = {k: sum(a.elapsed_time(b) for a, b in v) / len(v) for k, v in _events.items() if v}
    counts = {k: len(v) for k, v in _events.items() if v}
    _events.clear()
    return (out, counts) if with_counts else out


class stage:
    __slots__ = ("name", "a")

    def __init__(self, name):
        self.name = name
        self.a = None

    def __enter__(self):
        if _enabled and (_only is None or self.name in _only):
            self.a = torch.cuda.Event(enable_timing=True)
            self.a.record()
        return self

    def __exit__(self, *exc):
        if self.a is not None:
            b = torch.cuda.Event(enable_timing=True)
            b.record()
            _events[self.name].append((self.a, b))
        return False
