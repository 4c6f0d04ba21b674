"""FP8 global state: `fp8_autocast`, recipe bookkeeping, and the device-resident meta arenas.

Mirrors the behaviour of `transformer_engine.pytorch.fp8` that the reference relies on:
  * `fp8_autocast(enabled, fp8_recipe)` is re-entrant (te_llama.py:76,79 nest inside accelerate's outer
    autocast, utils/transformer_engine.py:129-134);
  * forward amaxes are reduced and ALL forward scales updated when the OUTERMOST autocast exits with
    grad enabled; backward scales are updated after the backward of the first FP8 module that ran in
    that outermost region (SURVEY.md Appendix A "Scale update");
  * with torch.distributed initialised and `recipe.reduce_amax`, the amax vector is MAX-all-reduced
    (one RCCL call per direction per step) before the update.

MI355X-first layout: instead of per-module meta tensors that are concatenated, updated and split
every step, all modules that share a recipe own slices of one arena ([H, CAP] amax history + [CAP]
scale / scale_inv / fp8_max, resident in HBM), so the whole model costs ONE `mi_scale_update` launch
(and one all-reduce) per direction per step.
"""
from __future__ import annotations

from contextlib import contextmanager
from typing import Dict, List, Optional, Tuple

import torch

from ..common.recipe import DelayedScaling, Format, MXFP8BlockScaling, Recipe, fmt_codes
from . import ops

__all__ = ["fp8_autocast", "FP8GlobalStateManager", "check_fp8_support", "check_mxfp8_support", "get_default_fp8_recipe"]


def check_fp8_support() -> Tuple[bool, str]:
    if not torch.cuda.is_available():
        return False, "no HIP device"
    from .. import _lib
    ok = _lib.load().mi_device_supported()
    return (True, "") if ok == 1 else (False, "device is not gfx950 (MI355X)")


def check_mxfp8_support() -> Tuple[bool, str]:
    """accelerate calls this before building MXFP8BlockScaling (utils/transformer_engine.py:157-159)."""
    return check_fp8_support()


def get_default_fp8_recipe() -> Recipe:
    return DelayedScaling()


class MetaArena:
    """Scaling state of every module slot under one (format, history, algo, margin, direction)."""

    CAP = 8192

    def __init__(self, key, device):
        fmt, H, algo, margin, forward = key
        self.key = key
        self.H, self.algo, self.margin, self.forward = H, algo, margin, forward
        self.device = device
        self.hist = torch.zeros((H, self.CAP), dtype=torch.float32, device=device)
        self.scale = torch.ones(self.CAP, dtype=torch.float32, device=device)
        self.scale_inv = torch.ones(self.CAP, dtype=torch.float32, device=device)
        self.fp8_max = torch.full((self.CAP,), fmt.value.max_fwd if forward else fmt.value.max_bwd,
                                  dtype=torch.float32, device=device)
        self.used = 0
        self.generation = 0  # bumped whenever scales may change (update, load_state): FP8 copies made before are stale
        self.reduce_amax = True
        self.group = None
        self._snap: Optional[torch.Tensor] = None  # copy of scale_inv taken at most once between two updates

    def alloc(self, n: int) -> int:
        if self.used + n > self.CAP:
            raise RuntimeError("FP8 meta arena exhausted; raise MetaArena.CAP")
        start = self.used
        self.used += n
        return start

    def reduce(self) -> None:
        """MAX-all-reduce this iteration's amax row over the data-parallel group: ONE collective for every slot of
        the arena (TE default `reduce_amax=True`, SURVEY.md 2.4 "FP8 amax all-reduce")."""
        n = self.used
        if n == 0 or not self.reduce_amax:
            return
        dist = torch.distributed
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1:
            dist.all_reduce(self.hist[0, :n], op=dist.ReduceOp.MAX, group=self.group)

    def snapshot(self) -> torch.Tensor:
        """scale_inv as of now, as ONE device copy shared by every module of the arena until the next update.  Autograd
        Functions keep views of it for their backward (the arena itself is rewritten at autocast exit, before backward)."""
        if self._snap is None or self._snap.numel() < self.used:
            self._snap = self.scale_inv[:self.CAP if self.used == 0 else self.used].clone()
        return self._snap

    def update(self) -> None:
        if self.used == 0:
            return
        n = self.used
        self._snap = None
        self.generation += 1
        self.reduce()
        ops.scale_update(self.hist[:, :n], self.scale[:n], self.scale_inv[:n], self.fp8_max[:n], self.margin, self.algo)


class ModuleMeta:
    """A module's window into an arena: `n` consecutive slots."""

    def __init__(self, arena: MetaArena, start: int, n: int):
        self.arena, self.start, self.n = arena, start, n

    def scale(self, i: int) -> torch.Tensor:
        return self.arena.scale[self.start + i:self.start + i + 1]

    def scale_inv(self, i: int) -> torch.Tensor:
        return self.arena.scale_inv[self.start + i:self.start + i + 1]

    def amax(self, i: int) -> torch.Tensor:
        return self.arena.hist[0, self.start + i:self.start + i + 1]

    def scale_inv_snapshot(self) -> torch.Tensor:
        """Read-only view [n] of the arena's snapshot (do not write into it: it is shared)."""
        return self.arena.snapshot()[self.start:self.start + self.n]

    def state(self) -> Dict[str, torch.Tensor]:
        a, s, n = self.arena, self.start, self.n
        return {"scale": a.scale[s:s + n].clone(), "scale_inv": a.scale_inv[s:s + n].clone(),
                "amax_history": a.hist[:, s:s + n].clone()}

    def load_state(self, st: Dict[str, torch.Tensor]) -> None:
        a, s, n = self.arena, self.start, self.n
        a._snap = None
        a.generation += 1
        a.scale[s:s + n].copy_(st["scale"])
        a.scale_inv[s:s + n].copy_(st["scale_inv"])
        h = st["amax_history"]
        if h.shape[0] != a.H:
            raise RuntimeError(f"amax history length {h.shape[0]} in checkpoint != recipe's {a.H}")
        a.hist[:, s:s + n].copy_(h)


class FP8GlobalStateManager:
    FP8_ENABLED = False
    FP8_CALIBRATION = False
    FP8_RECIPE: Optional[Recipe] = None
    FP8_GROUP = None
    FP8_AUTOCAST_DEPTH = 0
    IS_FIRST_FP8_MODULE = False
    _arenas: Dict[tuple, MetaArena] = {}

    @classmethod
    def reset(cls) -> None:
        cls.FP8_ENABLED = False
        cls.FP8_CALIBRATION = False
        cls.FP8_RECIPE = None
        cls.FP8_GROUP = None
        cls.FP8_AUTOCAST_DEPTH = 0
        cls.IS_FIRST_FP8_MODULE = False
        cls._arenas = {}

    @classmethod
    def is_fp8_enabled(cls) -> bool:
        return cls.FP8_ENABLED

    @classmethod
    def get_fp8_recipe(cls) -> Recipe:
        return cls.FP8_RECIPE if cls.FP8_RECIPE is not None else get_default_fp8_recipe()

    @classmethod
    def is_first_fp8_module(cls) -> bool:
        """True exactly once per outermost autocast region (its first FP8 module's backward runs last)."""
        tmp = cls.IS_FIRST_FP8_MODULE
        cls.IS_FIRST_FP8_MODULE = False
        return tmp

    @classmethod
    def arena(cls, recipe: DelayedScaling, forward: bool, device) -> MetaArena:
        key = (recipe.fp8_format, recipe.amax_history_len, recipe.amax_compute_algo, recipe.margin, forward)
        full = key + (str(device),)
        a = cls._arenas.get(full)
        if a is None:
            a = MetaArena(key, device)
            cls._arenas[full] = a
        a.reduce_amax = recipe.reduce_amax
        a.group = cls.FP8_GROUP
        return a

    @classmethod
    def reduce_and_update_fp8_tensors(cls, forward: bool) -> None:
        for a in cls._arenas.values():
            if a.forward == forward:
                a.update()

    @classmethod
    def fp8_autocast_enter(cls, enabled, calibrating, fp8_recipe, fp8_group):
        if enabled:
            ok, why = check_fp8_support()
            if not ok:
                raise RuntimeError(f"fp8_autocast(enabled=True): FP8 is not available: {why}")
        saved = (cls.FP8_ENABLED, cls.FP8_CALIBRATION, cls.FP8_RECIPE, cls.FP8_GROUP)
        cls.FP8_ENABLED = enabled
        cls.FP8_CALIBRATION = calibrating
        cls.FP8_RECIPE = fp8_recipe if fp8_recipe is not None else get_default_fp8_recipe()
        cls.FP8_GROUP = fp8_group
        if cls.FP8_AUTOCAST_DEPTH == 0:
            cls.IS_FIRST_FP8_MODULE = True
        cls.FP8_AUTOCAST_DEPTH += 1
        return saved

    @classmethod
    def fp8_autocast_exit(cls, enabled, saved) -> None:
        cls.FP8_ENABLED, cls.FP8_CALIBRATION, cls.FP8_RECIPE, cls.FP8_GROUP = saved
        cls.FP8_AUTOCAST_DEPTH -= 1
        if enabled and cls.FP8_AUTOCAST_DEPTH == 0 and torch.is_grad_enabled():
            cls.reduce_and_update_fp8_tensors(forward=True)


@contextmanager
def fp8_autocast(enabled: bool = True, calibrating: bool = False, fp8_recipe: Optional[Recipe] = None,
                 fp8_group=None):
    """Same call shape as te.pytorch.fp8_autocast (te_llama.py:76,79)."""
    saved = FP8GlobalStateManager.fp8_autocast_enter(enabled, calibrating, fp8_recipe, fp8_group)
    try:
        yield
    finally:
        FP8GlobalStateManager.fp8_autocast_exit(enabled, saved)
