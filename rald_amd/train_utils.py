"""Optimizer side of ``train_one_epoch`` (engine_generation.py:42-135) on flat
device storage.

The reference runs ``clip_grad_norm_`` -> ``torch.optim.AdamW.step`` -> ``update_ema`` over ~560 separate
parameter tensors (utils/misc.py:249-269, main_generation.py:161, engine_generation.py:29-40).  Here the
parameters, gradients, Adam moments and the EMA copy live in five flat fp32 buffers with one layout
(every tensor 16-byte aligned); ``param.data`` / ``param.grad`` are views into them, so module code and
checkpoints still see ordinary tensors, while the step itself is two streaming HIP launches
(``rald_optim_*``) and the data-parallel gradient exchange is a handful of large RCCL all-reduces on
slices of the flat gradient, launched as the backward pass finishes them (xGMI is per-link bound: few,
large messages).

Names follow what they replace: ``FlatAdamW`` has ``param_groups`` / ``step`` / ``zero_grad`` /
``state_dict`` like ``torch.optim.AdamW`` (so ``lr_sched.adjust_learning_rate`` works unchanged),
``clip_grad_norm_`` like ``torch.nn.utils``, ``update_ema(rate)`` like engine_generation.update_ema.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

from ._handles import _ptr, _stream
from ._lib import check, lib

ALIGN = 4            # elements: every tensor starts on a 16-byte boundary of the flat buffer


def flat_layout(shapes: Sequence[Sequence[int]]) -> Tuple[List[int], int]:
    """Offsets (in elements) of each tensor in the flat buffer, and the padded total."""
    offs, cur = [], 0
    for s in shapes:
        n = 1
        for d in s:
            n *= int(d)
        offs.append(cur)
        cur += -(-n // ALIGN) * ALIGN
    return offs, cur


class FlatAdamW:
    """``torch.optim.AdamW(params, lr)`` + gradient clipping + EMA on flat storage.

    After construction every ``p.data`` is a view of ``self.flat_p`` and ``p.grad`` a view of
    ``self.flat_g`` (zero-initialised), so a backward pass that accumulates into ``p.grad`` fills the
    flat gradient directly.  ``ema=True`` keeps the reference's ``ema_params`` list (deep copy of the
    parameters at construction, main_generation.py:147) as views of ``self.flat_ema``.
    """

    def __init__(self, params: Iterable[torch.nn.Parameter], lr: float = 1e-3, betas: Tuple[float, float] = (0.9, 0.999),
                 eps: float = 1e-8, weight_decay: float = 1e-2, ema: bool = False):
        self.params: List[torch.nn.Parameter] = [p for p in params]
        if not self.params:
            raise ValueError("optimizer got an empty parameter list")
        dev = self.params[0].device
        if dev.type != "cuda":
            raise RuntimeError("FlatAdamW runs on the HIP device only (no CPU fallback)")
        for p in self.params:
            if p.dtype != torch.float32 or p.device != dev:
                raise ValueError("all parameters must be float32 tensors on one device")
        self.offsets, self.numel = flat_layout([tuple(p.shape) for p in self.params])
        mk = lambda: torch.zeros(self.numel, device=dev, dtype=torch.float32)
        self.flat_p, self.flat_g, self.exp_avg, self.exp_avg_sq = mk(), mk(), mk(), mk()
        self.flat_ema = mk() if ema else None
        for p, o in zip(self.params, self.offsets):
            n = p.numel()
            self.flat_p[o:o + n].copy_(p.data.reshape(-1))
            p.data = self.flat_p[o:o + n].view(p.shape)
            p.grad = self.flat_g[o:o + n].view(p.shape)
        if ema:
            self.flat_ema.copy_(self.flat_p)
        self.ema_params = [self.flat_ema[o:o + p.numel()].view(p.shape) for p, o in zip(self.params, self.offsets)] if ema else None
        self.param_groups = [dict(params=self.params, lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay)]
        self.step_count = 0
        self._sumsq = torch.zeros(1, device=dev, dtype=torch.float64)
        self._norm_coef = torch.ones(2, device=dev, dtype=torch.float32)     # [total_norm, gradient factor]
        self._have_coef = False

    # -- torch.optim.Optimizer surface -------------------------------------------------------------
    def zero_grad(self, set_to_none: bool = False) -> None:
        self.flat_g.zero_()                      # grads stay views of the flat buffer (never set to None)

    def clip_grad_norm_(self, max_norm: float, pre_scale: float = 1.0) -> torch.Tensor:
        """``torch.nn.utils.clip_grad_norm_(parameters, max_norm)`` (utils/misc.py:262).  Returns the total
        norm as a 0-dim DEVICE tensor (the reference's ``norm``); the clipping itself is folded into the
        next ``step`` (no extra pass, no host sync).  ``pre_scale`` multiplies the gradient first - 1/world
        after a SUM all-reduce."""
        L = lib()
        check(L.rald_optim_grad_sumsq(C.c_void_p(_ptr(self.flat_g)), self.numel, C.c_void_p(_ptr(self._sumsq)), C.c_void_p(_stream())))
        check(L.rald_optim_clip_coef(C.c_void_p(_ptr(self._sumsq)), float(pre_scale), float(max_norm if max_norm else 0.0),
                                     C.c_void_p(_ptr(self._norm_coef)), C.c_void_p(_stream())))
        self._have_coef = True
        return self._norm_coef[0]

    def step(self, ema_rate: Optional[float] = None, write_back_grads: bool = False) -> None:
        """One AdamW step (torch defaults unless set in ``param_groups``).  ``ema_rate`` fuses
        ``update_ema(ema_params, model_params, rate)`` into the same pass."""
        g = self.param_groups[0]
        self.step_count += 1
        use_ema = ema_rate is not None
        if use_ema and self.flat_ema is None:
            raise RuntimeError("FlatAdamW was built with ema=False")
        check(lib().rald_optim_adamw_ema(
            C.c_void_p(_ptr(self.flat_p)), C.c_void_p(_ptr(self.flat_g)), C.c_void_p(_ptr(self.exp_avg)), C.c_void_p(_ptr(self.exp_avg_sq)),
            C.c_void_p(_ptr(self.flat_ema) if use_ema else 0), self.numel, C.c_void_p(_ptr(self._norm_coef) + 4 if self._have_coef else 0),
            float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]), float(g["weight_decay"]), self.step_count,
            float(ema_rate if use_ema else 0.0), int(write_back_grads), C.c_void_p(_stream())))
        self._have_coef = False
        self.mark_params_changed()

    def mark_params_changed(self) -> None:
        """The step rewrote every parameter through raw pointers into ``flat_p``; neither ``data_ptr()`` nor
        ``_version`` of a ``p`` moved (``p.data = view`` keeps p's own version counter).  Bump the counters so that
        everything keyed on them - the HIP handles' weight fingerprint (models_radar_generation._HipBacked),
        autograd's saved-tensor checks - sees the new weights."""
        torch._C._increment_version(self.params)

    def update_ema(self, rate: float = 0.99) -> None:
        """engine_generation.update_ema (:29-40) on the flat buffers (for iterations without a step)."""
        if self.flat_ema is None:
            raise RuntimeError("FlatAdamW was built with ema=False")
        check(lib().rald_optim_ema(C.c_void_p(_ptr(self.flat_ema)), C.c_void_p(_ptr(self.flat_p)), self.numel, float(rate), C.c_void_p(_stream())))

    # torch.optim.AdamW.state_dict() layout, so the reference's checkpoints ('optimizer' entry,
    # utils/misc.py:309-316) round-trip
    def state_dict(self) -> Dict[str, object]:
        state = {}
        for i, (p, o) in enumerate(zip(self.params, self.offsets)):
            n = p.numel()
            state[i] = dict(step=torch.tensor(float(self.step_count)), exp_avg=self.exp_avg[o:o + n].view(p.shape).clone(),
                            exp_avg_sq=self.exp_avg_sq[o:o + n].view(p.shape).clone())
        g = self.param_groups[0]
        group = {k: v for k, v in g.items() if k != "params"}
        group.update(amsgrad=False, maximize=False, foreach=None, capturable=False, differentiable=False, fused=None,
                     params=list(range(len(self.params))))
        return {"state": state if self.step_count else {}, "param_groups": [group]}

    def load_state_dict(self, sd: Dict[str, object]) -> None:
        g = sd["param_groups"][0]
        if len(g["params"]) != len(self.params):
            raise ValueError("loaded state dict has a different number of parameters")
        for k in ("lr", "betas", "eps", "weight_decay"):
            self.param_groups[0][k] = g[k]
        steps = set()
        for i, st in sd["state"].items():
            p, o = self.params[int(i)], self.offsets[int(i)]
            n = p.numel()
            self.exp_avg[o:o + n].copy_(st["exp_avg"].reshape(-1))
            self.exp_avg_sq[o:o + n].copy_(st["exp_avg_sq"].reshape(-1))
            steps.add(int(float(st["step"])))
        if len(steps) > 1:
            raise ValueError("per-parameter step counts differ: not representable on flat storage")
        self.step_count = steps.pop() if steps else 0


class GradReducer:
    """Data-parallel gradient exchange on the flat gradient (DistributedDataParallel's role,
    main_generation.py:157-159): SUM all-reduces of fixed-size slices, launched asynchronously as soon
    as the backward pass has finished every gradient in a slice.  The backward pass produces gradients
    from the LAST parameter to the first, so the finished region grows from the end of the buffer:
    ``mark_ready(lo)`` says "all elements >= lo are final".  The 1/world average is not applied here - it
    is folded into ``FlatAdamW.clip_grad_norm_(pre_scale=1/world)``.  One bucket ~ 64 MiB: on xGMI a ring
    all-reduce is bound per link, so a few large messages beat many small ones."""

    def __init__(self, flat_grad: torch.Tensor, bucket_bytes: int = 64 << 20, group=None):
        self.flat = flat_grad
        self.group = group
        n = flat_grad.numel()
        per = max(ALIGN, (bucket_bytes // 4) // ALIGN * ALIGN)
        # buckets are cut from the END (first to complete), so the possibly short one is at the front
        self.bounds: List[Tuple[int, int]] = []
        hi = n
        while hi > 0:
            lo = max(0, hi - per)
            self.bounds.append((lo, hi))
            hi = lo
        self._next = 0
        self._work = []
        self._done = [False] * len(self.bounds)
        self._covered: List[Tuple[int, int]] = []

    @property
    def world(self) -> int:
        return dist.get_world_size(self.group) if dist.is_available() and dist.is_initialized() else 1

    def start(self) -> None:
        self._next, self._work = 0, []
        self._done = [False] * len(self.bounds)
        self._covered = []

    def _launch(self, i: int) -> None:
        a, b = self.bounds[i]
        if self.world > 1:
            self._work.append(dist.all_reduce(self.flat[a:b], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        self._done[i] = True

    def mark_range(self, lo: int, hi: int) -> int:
        """Elements [lo, hi) of the flat gradient are final (a transformer block's parameters, the encoder's ...): every
        not-yet-launched bucket that now lies entirely inside finished territory is launched.  The backward pass of EdmTrainer
        finishes the blocks from the last to the first and the radar encoder (which sits at the END of the flat buffer) last,
        so the finished region is a union of intervals, not a suffix.  Returns how many buckets were launched."""
        ivs = sorted(self._covered + [(int(lo), int(hi))])
        merged: List[Tuple[int, int]] = []
        for a, b in ivs:
            if merged and a <= merged[-1][1]:
                merged[-1] = (merged[-1][0], max(merged[-1][1], b))
            else:
                merged.append((a, b))
        self._covered = merged
        launched = 0
        for i, (a, b) in enumerate(self.bounds):
            if not self._done[i] and any(ca <= a and b <= cb for ca, cb in merged):
                self._launch(i)
                launched += 1
        return launched

    def mark_ready(self, lo: int) -> int:
        """Launch every not-yet-launched bucket that lies entirely at or above element ``lo``;
        returns how many were launched."""
        return self.mark_range(lo, self.flat.numel())

    def finish(self) -> float:
        """Launch what is left, wait for everything; returns the pre_scale (1/world) for the optimizer."""
        for i in range(len(self.bounds)):
            if not self._done[i]:
                self._launch(i)
        for w in self._work:
            w.wait()
        self._work = []
        return 1.0 / self.world
