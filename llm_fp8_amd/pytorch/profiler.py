"""Live per-kernel timing with HIP events on the launch stream (used by bench.py's `roofline` object).
The hot-path wrappers in ops.py call `KernelTimer.active.span(...)` when a timer is installed."""
from __future__ import annotations

from collections import defaultdict
from contextlib import contextmanager
from typing import Dict, List, Optional, Tuple

import torch


class KernelTimer:
    active: Optional["KernelTimer"] = None

    def __init__(self):
        self.spans: List[Tuple[str, float, float, torch.cuda.Event, torch.cuda.Event]] = []

    @contextmanager
    def span(self, kind: str, tag: str, work: float, bytes_: float = 0.0):
        s = torch.cuda.Event(enable_timing=True)
        e = torch.cuda.Event(enable_timing=True)
        s.record()
        try:
            yield
        finally:
            e.record()
            self.spans.append((kind, tag, work, bytes_, s, e))

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
