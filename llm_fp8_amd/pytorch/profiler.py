"""Live per-kernel timing with HIP events on the launch stream (used by bench.py's `roofline` object).
The hot-path wrappers in ops.py call `KernelTimer.active.span(...)` when a timer is installed."""
from __future__ import annotations

from collections import defaultdict
from contextlib import contextmanager
from typing import Dict, List, Optional, Tuple

import torch


class _NullSpan:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NULL = _NullSpan()


class _Span:
    __slots__ = ("timer", "rec")

    def __init__(self, timer, rec):
        self.timer, self.rec = timer, rec

    def __enter__(self):
        self.rec[4].record()
        return None

    def __exit__(self, *exc):
        self.rec[5].record()
        self.timer.spans.append(self.rec)
        return False


class KernelTimer:
    """`kinds`: the span kinds to bracket (None = all).  Each bracketed launch costs two event records on the stream
    (measured: ~4.5 us of queue time per event on MI355X), so bench.py brackets only the GEMMs inside its timed region."""
    active: Optional["KernelTimer"] = None

    def __init__(self, kinds=None):
        self.spans: List[tuple] = []
        self.kinds = None if kinds is None else frozenset(kinds)

    def span(self, kind: str, tag: str, work: float, bytes_: float = 0.0):
        if self.kinds is not None and kind not in self.kinds:
            return _NULL
        return _Span(self, (kind, tag, work, bytes_, torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)))

    @contextmanager
    def install(self):
        prev, KernelTimer.active = KernelTimer.active, self
        try:
            yield self
        finally:
            KernelTimer.active = prev

    def summarize(self) -> Dict[str, dict]:
        """{kind: {launches, seconds, work, bytes, by_tag: {tag: {...}}}} -- call after a device sync."""
        out: Dict[str, dict] = {}
        for kind, tag, work, nbytes, s, e in self.spans:
            t = s.elapsed_time(e) * 1e-3
            k = out.setdefault(kind, {"launches": 0, "seconds": 0.0, "work": 0.0, "bytes": 0.0, "by_tag": {}})
            k["launches"] += 1; k["seconds"] += t; k["work"] += work; k["bytes"] += nbytes
            g = k["by_tag"].setdefault(tag, {"launches": 0, "seconds": 0.0, "work": 0.0})
            g["launches"] += 1; g["seconds"] += t; g["work"] += work
        return out
