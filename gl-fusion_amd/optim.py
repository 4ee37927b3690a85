"""Optimizer of the reference's training loop (GLfusion/main.py:158-169) on the HIP engine.

`Adam` has torch.optim.Adam's constructor, param_groups and state layout ('step', 'exp_avg', 'exp_avg_sq' per
parameter -- optimizer checkpoints are interchangeable with torch.optim.Adam's), so
`torch.optim.lr_scheduler.CosineAnnealingLR` (main.py:168) drives it unchanged; `step()` updates every parameter
that has a gradient with ONE kernel launch per (param group, step count) (glf_adam_step) instead of ~1 500 small
ATen kernels.  Parameters without a gradient are skipped and get no state, exactly as in torch (the dead
`network.*` template and `align_channel` never receive one on this path).
"""
from __future__ import annotations

from typing import Dict, List

import numpy as np
import torch

from ._lib import check, lib
from .ops import _p, _stream, refresh_weights

CHUNK = 1 << 16          # elements per table row (one workgroup pass)


def _chunk_rows(entries: List[tuple]) -> np.ndarray:
    """(p_ptr, g_ptr, m_ptr, v_ptr, n) per parameter -> int64 table with one row per CHUNK elements."""
    a = np.asarray(entries, dtype=np.int64).reshape(-1, 5)
    n = a[:, 4]
    k = (n + CHUNK - 1) // CHUNK
    idx = np.repeat(np.arange(len(a)), k)
    first = np.cumsum(k) - k
    off = (np.arange(int(k.sum())) - np.repeat(first, k)) * CHUNK
    rows = a[idx].copy()
    rows[:, 0:4] += (off * 4)[:, None]
    rows[:, 4] = np.minimum(CHUNK, n[idx] - off)
    return rows


class _Table:
    """Device copy of a pointer table, re-uploaded only when a pointer changed (gradients that live in the
    all-reduce buckets never move; freshly allocated ones usually come back at the same addresses).  Uploads go
    through two alternating pinned host buffers so that step() never synchronises the host with the GPU."""

    def __init__(self):
        self.key = None
        self.dev = None
        self.host = [None, None]
        self.events = [None, None]
        self.turn = 0

    def get(self, entries: List[tuple], device) -> torch.Tensor:
        key = tuple(entries)
        if key == self.key:
            return self.dev
        rows = _chunk_rows(entries)
        i = self.turn
        self.turn ^= 1
        if self.host[i] is None or self.host[i].shape[0] < rows.shape[0]:
            self.host[i] = torch.empty(max(rows.shape[0], 64), 5, dtype=torch.int64).pin_memory()
            self.events[i] = None
        if self.events[i] is not None:
            self.events[i].synchronize()                    # the copy issued two uploads ago; long finished
        self.host[i][:rows.shape[0]].copy_(torch.from_numpy(rows))
        if self.dev is None or self.dev.shape[0] != rows.shape[0] or self.dev.device != device:
            self.dev = torch.empty(rows.shape[0], 5, dtype=torch.int64, device=device)
        self.dev.copy_(self.host[i][:rows.shape[0]], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.events[i] = ev
        self.key = key
        return self.dev


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0,
                 amsgrad: bool = False):
        if lr < 0.0 or eps < 0.0 or weight_decay < 0.0 or not (0.0 <= betas[0] < 1.0) or not (0.0 <= betas[1] < 1.0):
            raise ValueError("Invalid Adam hyper-parameter")           # torch.optim.Adam raises ValueError too
        if amsgrad:
            raise NotImplementedError("glfusion_amd.optim.Adam: amsgrad is not used by the reference (main.py:162) and not built")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False))
        self._tables: Dict[tuple, _Table] = {}

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for gi, group in enumerate(self.param_groups):
            b1, b2 = group["betas"]
            lr = group["lr"]
            by_step: Dict[int, List[tuple]] = {}
            dev = None
            updated = []
            for p in group["params"]:
                g = p.grad
                if g is None:
                    continue
                if g.is_sparse:
                    raise RuntimeError("Adam does not support sparse gradients, please consider SparseAdam instead")
                if not p.is_cuda or p.dtype != torch.float32 or g.dtype != torch.float32:
                    raise RuntimeError("glfusion_amd.optim.Adam: parameters and gradients must be CUDA(HIP) float32 tensors "
                                       "(the engine has no CPU fallback)")
                if not p.is_contiguous():
                    raise RuntimeError("glfusion_amd.optim.Adam: non-contiguous parameter")
                if not g.is_contiguous():
                    g = g.contiguous()
                    p.grad = g
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                t = int(st["step"]) + 1
                st["step"] = st["step"].new_tensor(float(t)) if isinstance(st["step"], torch.Tensor) else t
                by_step.setdefault(t, []).append((p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(),
                                                  st["exp_avg_sq"].data_ptr(), p.numel()))
                dev = p.device
                updated.append(p)
            for t, entries in by_step.items():
                table = self._tables.setdefault((gi, len(by_step) > 1 and t), _Table()).get(entries, dev)
                check(lib.glf_adam_step(_p(table), table.shape[0], float(lr), float(b1), float(b2), float(group["eps"]),
                                        float(group["weight_decay"]), t, _stream()), "adam_step")
            if updated:
                # the kernel writes through raw pointers: tell autograd (and every cache keyed on `_version`: the
                # tap-major / transposed / pre-split weight layouts and the measured maxima in ops.py) that these changed
                torch.autograd.graph.increment_version(updated)
        # ... and rebuild every registered weight-derived image in four launches (instead of ~5 launches per conv when the
        # next forward finds its caches stale)
        refresh_weights()
        return loss
