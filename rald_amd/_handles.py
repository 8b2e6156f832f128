"""Python owners of the C handles: device-memory plumbing (torch tensors -> raw pointers, the
current torch stream -> hipStream_t) around librald_hip.so.  No arithmetic happens here."""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Iterable, Optional, Tuple

import torch

from . import _lib
from ._lib import DitConfig, check, lib


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: torch.Tensor) -> int:
    return t.data_ptr()


def _need_cuda(t: torch.Tensor, what: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(f"rald_amd: {what} must live on the GPU (cuda tensor); this package has no CPU path")


def _f32c(t: torch.Tensor) -> torch.Tensor:
    return t.detach().to(torch.float32).contiguous()


# Launch-bound regime (small batches): one sampler run is ~10 000 kernel launches of a few
# microseconds each, so the whole call is captured once into a hipGraph (torch.cuda.CUDAGraph on
# the current stream - the library only enqueues kernels on the stream it is given) and replayed.
# RALD_GRAPH=0 disables it; batches above GRAPH_MAX_BATCH run eagerly (they are GPU-bound).
GRAPH_MAX_BATCH = int(os.environ.get("RALD_GRAPH_MAX_BATCH", "16"))


def _graphs_enabled() -> bool:
    return os.environ.get("RALD_GRAPH", "1") != "0"


class _GraphCache:
    """key -> (CUDAGraph, static inputs, static outputs, workspace generation).  Everything a captured graph points at
    (handle workspace, sigma table, static tensors) must stay put.  The library may reallocate its workspace inside ANY
    call (a larger batch through denoise / encode_cond / forward), so every replay first compares the handle's
    workspace generation (rald_*_workspace_generation) with the one recorded after capture and re-captures on a
    mismatch; owners additionally call clear() when something on the Python side changes."""

    def __init__(self, generation=lambda: 0):
        self.entries: Dict[tuple, tuple] = {}
        self.generation = generation

    def clear(self):
        self.entries.clear()

    def run(self, key, inputs, make_outputs, fn):
        ent = self.entries.get(key)
        if ent is not None and ent[3] != self.generation():      # the workspace moved since capture: the graph is stale
            del self.entries[key]
            ent = None
        if ent is None:
            static_in = [t.clone() for t in inputs]
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):                   # warm-up outside capture: lazy allocations,
                outs = make_outputs()                       # function attributes, sigma tables
                fn(static_in, outs)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                outs = make_outputs()
                fn(static_in, outs)
            ent = (g, static_in, outs, self.generation())
            self.entries[key] = ent
        g, static_in, outs, _ = ent
        for dst, src in zip(static_in, inputs):
            dst.copy_(src, non_blocking=True)
        g.replay()
        return [o.clone() for o in outs]


class DitHandle:
    """rald_dit* + its condition cache.  One handle per module per device."""

    def __init__(self, cfg: DitConfig):
        self.cfg = cfg
        self._h = C.c_void_p()
        check(lib().rald_dit_create(C.byref(cfg), C.byref(self._h)))
        self._graphs = _GraphCache(lambda: lib().rald_dit_workspace_generation(self._h))
        self._reserved = 0
        self._sched = None
        # Two-stream schedule of an NFE between 128 and 255 samples: which of the two bit-identical schedules is faster depends on the box
        # (clocks under MFMA load differ by ~6 % across the pool; two interleaved half-batches measured +2.5 % on slower boxes and -1.5 % on the
        # fastest).  Opt-in: with autotune_two_stream = True the first NFE of such a batch times both (eight NFEs each way, once per handle
        # and batch size) and keeps the winner; off by default - the library's fixed threshold (256) then decides, and a profile of the process
        # shows one schedule only.  An explicit set_two_stream_min_batch() also turns it off.
        self.autotune_two_stream = False
        self._two_stream_tuned = {}

    def __del__(self):
        try:
            if self._h:
                lib().rald_dit_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass

    def load(self, named: Iterable[Tuple[str, torch.Tensor]]) -> None:
        for name, t in named:
            t = _f32c(t)                       # host or device fp32; the library stages either
            check(lib().rald_dit_load_weight(self._h, name.encode(), C.c_void_p(_ptr(t)), t.numel()))
        check(lib().rald_dit_finalize(self._h))

    def set_sigmas(self, sigmas) -> None:
        s = [float(v) for v in sigmas]
        arr = (C.c_float * len(s))(*s)
        check(lib().rald_dit_set_sigmas(self._h, arr, len(s), C.c_void_p(_stream())))

    def new_cache(self, batch: int, device) -> torch.Tensor:
        nbytes = lib().rald_dit_cond_cache_bytes(self._h, batch)
        return torch.empty(nbytes, dtype=torch.uint8, device=device)

    def encode_cond_tokens(self, tokens: torch.Tensor) -> torch.Tensor:
        _need_cuda(tokens, "condition tokens")
        tokens = _f32c(tokens)
        B, T, Cd = tokens.shape
        if T != self.cfg.n_cond_tokens or Cd != self.cfg.context_dim:
            raise RuntimeError(f"condition tokens must be [B,{self.cfg.n_cond_tokens},{self.cfg.context_dim}], got {tuple(tokens.shape)}")
        cache = self.new_cache(B, tokens.device)
        check(lib().rald_dit_encode_cond_tokens(self._h, C.c_void_p(_ptr(tokens)), B, C.c_void_p(_ptr(cache)), C.c_void_p(_stream())))
        return cache

    def encode_cond(self, cube: torch.Tensor, want_tokens: bool = True):
        _need_cuda(cube, "radar cube")
        cube = _f32c(cube)
        B = cube.shape[0]
        cache = self.new_cache(B, cube.device)
        tokens = None
        tp = C.c_void_p(0)
        if want_tokens:
            tokens = torch.empty(B, self.cfg.n_cond_tokens, self.cfg.n_heads * self.cfg.d_head, device=cube.device, dtype=torch.float32)
            tp = C.c_void_p(_ptr(tokens))
        check(lib().rald_dit_encode_cond(self._h, C.c_void_p(_ptr(cube)), B, tp, C.c_void_p(_ptr(cache)), C.c_void_p(_stream())))
        return tokens, cache

    def denoise(self, x: torch.Tensor, cache: torch.Tensor, sigma_row: int = 0, per_sample: bool = False,
                raw_F: bool = False) -> torch.Tensor:
        _need_cuda(x, "x")
        x = _f32c(x)
        if x.dim() != 3 or x.shape[1] != self.cfg.n_latents or x.shape[2] != self.cfg.channels:
            raise RuntimeError(f"x must be [B,{self.cfg.n_latents},{self.cfg.channels}], got {tuple(x.shape)}")
        if cache.dtype != torch.uint8 or not cache.is_cuda or cache.numel() != lib().rald_dit_cond_cache_bytes(self._h, x.shape[0]):
            raise RuntimeError(f"condition cache of {cache.numel()} bytes does not belong to a batch of {x.shape[0]} "
                               f"(expected {lib().rald_dit_cond_cache_bytes(self._h, x.shape[0])} bytes): encode the condition for the same batch")
        out = torch.empty_like(x)
        B = x.shape[0]
        if self.autotune_two_stream and 128 <= B < 256 and B not in self._two_stream_tuned and not torch.cuda.is_current_stream_capturing():
            self._tune_two_stream(x, cache, sigma_row, per_sample, raw_F, out)
        check(lib().rald_dit_denoise(self._h, C.c_void_p(_ptr(x)), B, sigma_row, int(per_sample),
                                     C.c_void_p(_ptr(cache)), C.c_void_p(_ptr(out)), int(raw_F), C.c_void_p(_stream())))
        return out

    def _tune_two_stream(self, x, cache, sigma_row, per_sample, raw_F, out) -> None:
        """Times the whole-batch and the two-half-batch schedule of this NFE (same inputs, same bits out) and sets the library's threshold."""
        B = x.shape[0]
        prev = lib().rald_dit_two_stream_min_batch(self._h)

        def run(n):
            for _ in range(n):
                check(lib().rald_dit_denoise(self._h, C.c_void_p(_ptr(x)), B, sigma_row, int(per_sample), C.c_void_p(_ptr(cache)),
                                             C.c_void_p(_ptr(out)), int(raw_F), C.c_void_p(_stream())))
        ms = {}
        for mode, mb in (("whole", 0), ("split", B)):
            self._set_two_stream(mb)
            run(2)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            run(6)
            e1.record()
            e1.synchronize()
            ms[mode] = e0.elapsed_time(e1)
        split = ms["split"] < 0.995 * ms["whole"]
        self._two_stream_tuned[B] = (split, ms["whole"] / 6, ms["split"] / 6)
        self._set_two_stream(min(prev if prev > 0 else 256, B) if split else max(prev, B + 1))

    def _set_two_stream(self, min_batch: int) -> None:
        self._graphs.clear()                           # the workspace is re-planned
        check(lib().rald_dit_set_two_stream_min_batch(self._h, int(min_batch)))

    def set_two_stream_min_batch(self, min_batch: int) -> None:
        """From `min_batch` samples up an NFE runs as two half-batches on two HIP streams (default 256; 0 = never).  Setting it by hand
        switches the per-box choice for batches of 128-255 (see __init__) off."""
        self.autotune_two_stream = False
        self._set_two_stream(min_batch)

    def profile_begin(self) -> None:
        check(lib().rald_dit_profile_begin(self._h))

    def profile_set_kinds(self, mask: int) -> None:
        """Which kinds (bit k, see profile_end_kinds) are bracketed from now on; profile_begin resets to all four."""
        check(lib().rald_dit_profile_set_kinds(self._h, int(mask)))

    def profile_end(self):
        ms, n = C.c_double(0), C.c_int32(0)
        check(lib().rald_dit_profile_end(self._h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def profile_end_kinds(self):
        """[(ms, launches)] x 4: FF1 GEGLU GEMM, attn1 / attn2 output projection + residual + AdaLN (K = 512), FF2 + residual + AdaLN."""
        ms, n = (C.c_double * 4)(), (C.c_int32 * 4)()
        check(lib().rald_dit_profile_end_kinds(self._h, ms, n))
        return [(ms[i], n[i]) for i in range(4)]

    def reserve(self, batch: int) -> None:
        if batch > self._reserved:
            self._graphs.clear()                       # workspace may move: captured graphs are stale
            check(lib().rald_dit_reserve(self._h, batch))
            self._reserved = batch

    def _sample_eager(self, latents, cache, out, num_steps, sigma_min, sigma_max, rho):
        check(lib().rald_dit_sample(self._h, C.c_void_p(_ptr(latents)), latents.shape[0], C.c_void_p(_ptr(cache)), num_steps,
                                    sigma_min, sigma_max, rho, C.c_void_p(_ptr(out)), C.c_void_p(_stream())))

    def sample(self, latents: torch.Tensor, cache: torch.Tensor, num_steps: int = 18, sigma_min: float = 0.002,
               sigma_max: float = 80.0, rho: float = 7.0, use_graph: Optional[bool] = None) -> torch.Tensor:
        _need_cuda(latents, "latents")
        latents = _f32c(latents)
        B = latents.shape[0]
        if cache.dtype != torch.uint8 or not cache.is_cuda or cache.numel() != lib().rald_dit_cond_cache_bytes(self._h, B):
            raise RuntimeError(f"condition cache of {cache.numel()} bytes does not belong to a batch of {B}: encode the condition for the same batch")
        self.reserve(B)
        sched = (num_steps, float(sigma_min), float(sigma_max), float(rho))
        if sched != self._sched:                       # the sampler's sigma table is rebuilt for a new schedule
            self._graphs.clear()
            self._sched = sched
        if use_graph is None:
            use_graph = _graphs_enabled() and B <= GRAPH_MAX_BATCH
        if not use_graph:
            out = torch.empty_like(latents)
            self._sample_eager(latents, cache, out, *sched)
            return out
        (out,) = self._graphs.run(("sample", B) + sched, [latents, cache], lambda: [torch.empty_like(latents)],
                                  lambda ins, outs: self._sample_eager(ins[0], ins[1], outs[0], *sched))
        return out


# ---- kernel-level wrappers used by the parity tests and microbenchmarks -----------------------
def op_gemm_nt(A: torch.Tensor, B: torch.Tensor, bias: Optional[torch.Tensor] = None, epilogue: int = 0,
               C_inout: Optional[torch.Tensor] = None, alpha: float = 1.0) -> torch.Tensor:
    """A [batch?,M,K] bf16, B [batch?,N,K] bf16 -> C.  epilogue 0 bf16, 1 f32, 2 f32 accumulate into
    C_inout, 3 GEGLU (B rows / bias pre-packed), 4 softmax over aligned groups of 64 columns (exp2 units, bf16), 5 fp16 slab =
    2^-6 x the product, saturating (the split-K partial sums of the small-batch path)."""
    batched = A.dim() == 3 or B.dim() == 3
    batch = (A.shape[0] if A.dim() == 3 else B.shape[0]) if batched else 1
    M, K = A.shape[-2], A.shape[-1]
    N = B.shape[-2]
    sA = A.stride(0) if A.dim() == 3 else 0
    sB = B.stride(0) if B.dim() == 3 else 0
    nc = N // 2 if epilogue == 3 else N
    if epilogue == 2:
        out = C_inout
    else:
        shape = (batch, M, nc) if batched else (M, nc)
        out = torch.empty(shape, device=A.device, dtype=torch.float16 if epilogue == 5 else (torch.bfloat16 if epilogue in (0, 3, 4) else torch.float32))
    sC = out.stride(0) if out.dim() == 3 else 0
    check(lib().rald_op_gemm_nt(C.c_void_p(_ptr(A)), A.stride(-2), sA, C.c_void_p(_ptr(B)), B.stride(-2), sB,
                                C.c_void_p(_ptr(out)), out.stride(-2), sC,
                                C.c_void_p(_ptr(bias) if bias is not None else 0), M, N, K, batch, alpha, epilogue,
                                C.c_void_p(_stream())))
    return out


def op_gemm_tn(A: torch.Tensor, B: torch.Tensor, C_inout: torch.Tensor, colsum: Optional[torch.Tensor] = None, atomics: bool = True) -> torch.Tensor:
    """C_inout [N1,N2] f32 += A^T.B for row-major bf16 A [M,N1], B [M,N2] (column slices allowed); colsum [N1] f32 += column sums of A.
    atomics=False: the row ranges meet in a workspace and are added in order by a second launch (bit-reproducible)."""
    M, N1 = A.shape
    N2 = B.shape[1]
    assert A.dtype == torch.bfloat16 and B.dtype == torch.bfloat16 and C_inout.dtype == torch.float32 and B.shape[0] == M
    assert A.stride(1) == 1 and B.stride(1) == 1 and C_inout.stride(1) == 1 and C_inout.shape == (N1, N2)
    if atomics:
        check(lib().rald_op_gemm_tn(C.c_void_p(_ptr(A)), A.stride(0), C.c_void_p(_ptr(B)), B.stride(0), C.c_void_p(_ptr(C_inout)), C_inout.stride(0),
                                    C.c_void_p(_ptr(colsum) if colsum is not None else 0), M, N1, N2, C.c_void_p(_stream())))
        return C_inout
    nbytes = lib().rald_op_gemm_tn_workspace_bytes(M, N1, N2)
    ws = torch.empty(max(nbytes, 16), device=A.device, dtype=torch.uint8)
    check(lib().rald_op_gemm_tn_ws(C.c_void_p(_ptr(A)), A.stride(0), C.c_void_p(_ptr(B)), B.stride(0), C.c_void_p(_ptr(C_inout)), C_inout.stride(0),
                                   C.c_void_p(_ptr(colsum) if colsum is not None else 0), M, N1, N2, C.c_void_p(_ptr(ws)), nbytes, C.c_void_p(_stream())))
    return C_inout


def op_gemm_resid_ln(A: torch.Tensor, W: torch.Tensor, bias: torch.Tensor, x: torch.Tensor, g: torch.Tensor, b: torch.Tensor,
                     gstride: int = 0, rows_per_group: int = 1 << 30, add_one: float = 0.0, eps: float = 1e-5) -> torch.Tensor:
    """x [M,512] f32 += A [M,K] bf16 @ W [512,K]^T + bias (in place); returns h = LN(x)*(add_one+g)+b as bf16."""
    M, K = A.shape
    h = torch.empty(M, 512, device=A.device, dtype=torch.bfloat16)
    check(lib().rald_op_gemm_resid_ln(C.c_void_p(_ptr(A)), A.stride(0), C.c_void_p(_ptr(W)), W.stride(0), C.c_void_p(_ptr(bias)),
                                      C.c_void_p(_ptr(x)), C.c_void_p(_ptr(h)), C.c_void_p(_ptr(g)), C.c_void_p(_ptr(b)), gstride,
                                      rows_per_group, add_one, eps, M, K, C.c_void_p(_stream())))
    return h


def op_layernorm(x: torch.Tensor, g: torch.Tensor, b: torch.Tensor, gstride: int = 0, rows_per_group: int = 1,
                 add_one: float = 0.0, eps: float = 1e-5) -> torch.Tensor:
    M, D = x.shape
    out = torch.empty(M, D, device=x.device, dtype=torch.bfloat16)
    check(lib().rald_op_layernorm(C.c_void_p(_ptr(x)), C.c_void_p(_ptr(out)), M, D, C.c_void_p(_ptr(g)), C.c_void_p(_ptr(b)),
                                  gstride, rows_per_group, add_one, eps, C.c_void_p(_stream())))
    return out


def op_attention(Q: torch.Tensor, K: torch.Tensor, Vt: torch.Tensor, nk: int, heads: int, scale: float) -> torch.Tensor:
    """Q [B,nq,H*64], K [B,k_rows,H*64], Vt [B,H*64,ldvt] (bf16) -> O [B,nq,H*64] bf16."""
    Bn, nq, HD = Q.shape
    O = torch.empty(Bn, nq, HD, device=Q.device, dtype=torch.bfloat16)
    check(lib().rald_op_attention(C.c_void_p(_ptr(Q)), Q.stride(1), Q.stride(0), C.c_void_p(_ptr(K)), K.stride(1), K.stride(0),
                                  C.c_void_p(_ptr(Vt)), Vt.stride(1), Vt.stride(0), C.c_void_p(_ptr(O)), O.stride(1), O.stride(0),
                                  nq, nk, K.shape[1], heads, Bn, scale, C.c_void_p(_stream())))
    return O


def op_attention_split(Q: torch.Tensor, K: torch.Tensor, Vt: torch.Tensor, nk: int, heads: int, scale: float, ksplit: int) -> torch.Tensor:
    """op_attention with the keys split over `ksplit` workgroups per query block (+ combine pass)."""
    Bn, nq, HD = Q.shape
    O = torch.empty(Bn, nq, HD, device=Q.device, dtype=torch.bfloat16)
    scratch = torch.empty(max(8, lib().rald_op_attention_split_scratch_bytes(max(ksplit, 32), nq, heads, Bn)), device=Q.device, dtype=torch.uint8)
    check(lib().rald_op_attention_split(C.c_void_p(_ptr(Q)), Q.stride(1), Q.stride(0), C.c_void_p(_ptr(K)), K.stride(1), K.stride(0),
                                        C.c_void_p(_ptr(Vt)), Vt.stride(1), Vt.stride(0), C.c_void_p(_ptr(O)), O.stride(1), O.stride(0),
                                        nq, nk, K.shape[1], heads, Bn, scale, ksplit, C.c_void_p(_ptr(scratch)), C.c_void_p(_stream())))
    return O


def op_attention_vrow(Q: torch.Tensor, K: torch.Tensor, V: torch.Tensor, heads: int, scale: float) -> torch.Tensor:
    """Q [B,nq,H*64], K / V [B,nk,H*64] (bf16, possibly column slices of one fused buffer) -> O [B,nq,H*64] bf16."""
    Bn, nq, HD = Q.shape
    O = torch.empty(Bn, nq, HD, device=Q.device, dtype=torch.bfloat16)
    check(lib().rald_op_attention_vrow(C.c_void_p(_ptr(Q)), Q.stride(1), Q.stride(0), C.c_void_p(_ptr(K)), K.stride(1), K.stride(0),
                                       C.c_void_p(_ptr(V)), V.stride(1), V.stride(0), C.c_void_p(_ptr(O)), O.stride(1), O.stride(0),
                                       nq, K.shape[1], heads, Bn, scale, C.c_void_p(_stream())))
    return O


def op_attention_f16kv(Q: torch.Tensor, KV: torch.Tensor, nk: int, heads: int, ksplit: int = -1, shared_q: bool = False) -> torch.Tensor:
    """Folded-encoder attention: Q fp32 [B,nq,H*64] ([nq,H*64] with shared_q) already times scale*log2(e); KV fp16 [B,k_rows,64] = key AND
    value row of every head (rows nk.. must be zero) -> O bf16 [B,nq,H*64].  ksplit: -1 pick, 0/1 off, > 1 workgroups per query block."""
    Bn, k_rows = KV.shape[0], KV.shape[1]
    nq, HD = Q.shape[-2], Q.shape[-1]
    assert Q.dtype == torch.float32 and KV.dtype == torch.float16 and Q.is_contiguous() and KV.is_contiguous() and HD == heads * 64
    O = torch.empty(Bn, nq, HD, device=KV.device, dtype=torch.bfloat16)
    scratch = torch.empty(lib().rald_op_attention_split_scratch_bytes(max(ksplit, 32), nq, heads, Bn) // 4, device=KV.device, dtype=torch.float32)
    check(lib().rald_op_attention_f16kv(C.c_void_p(_ptr(Q)), HD, 0 if shared_q else nq * HD, C.c_void_p(_ptr(KV)), C.c_void_p(_ptr(O)), HD, nq * HD,
                                        nq, nk, k_rows, heads, Bn, ksplit, C.c_void_p(_ptr(scratch)), C.c_void_p(_stream())))
    return O


def op_ae_enc_features(pc: torch.Tensor, basis: torch.Tensor, var_factor: torch.Tensor):
    """pc [B,P,3] fp32, basis [3,24], var_factor [52,52] -> (F, G) fp16 [B, round_up(P,64), 64] (rald_amd/csrc/ae_encode.hip)."""
    Bn, P = pc.shape[0], pc.shape[1]
    Pp = (P + 63) // 64 * 64
    F = torch.empty(Bn, Pp, 64, device=pc.device, dtype=torch.float16)
    G = torch.empty_like(F)
    check(lib().rald_op_ae_enc_features(C.c_void_p(_ptr(_f32c(pc))), C.c_void_p(_ptr(_f32c(basis))), C.c_void_p(_ptr(_f32c(var_factor))),
                                        C.c_void_p(_ptr(F)), C.c_void_p(_ptr(G)), Bn, P, Pp, C.c_void_p(_stream())))
    return F, G


class AeHandle:
    """rald_ae*: encode / decode_latents / decode_queries of the set-latent autoencoder."""

    def __init__(self, cfg):
        self.cfg = cfg
        self._h = C.c_void_p()
        check(lib().rald_ae_create(C.byref(cfg), C.byref(self._h)))
        self._graphs = _GraphCache(lambda: lib().rald_ae_workspace_generation(self._h))
        self._dec_batch = 0

    def __del__(self):
        try:
            if self._h:
                lib().rald_ae_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass

    def load(self, named) -> None:
        for name, t in named:
            t = _f32c(t)
            check(lib().rald_ae_load_weight(self._h, name.encode(), C.c_void_p(_ptr(t)), t.numel()))
        check(lib().rald_ae_finalize(self._h))

    def encode(self, pc: torch.Tensor, eps: torch.Tensor, want_moments: bool = False):
        _need_cuda(pc, "point cloud")
        pc, eps = _f32c(pc), _f32c(eps).to(pc.device)
        B = pc.shape[0]
        M, L = self.cfg.num_latents, self.cfg.latent_dim
        if pc.shape[1] != self.cfg.num_inputs:
            raise AssertionError(f"encode expects {self.cfg.num_inputs} points, got {pc.shape[1]}")   # models_ae.py:354
        z = torch.empty(B, M, L, device=pc.device, dtype=torch.float32)
        kl = torch.empty(B, device=pc.device, dtype=torch.float32)
        mean = torch.empty_like(z) if want_moments else None
        logvar = torch.empty_like(z) if want_moments else None
        check(lib().rald_ae_encode(self._h, C.c_void_p(_ptr(pc)), B, C.c_void_p(_ptr(eps)),
                                   C.c_void_p(_ptr(mean) if want_moments else 0), C.c_void_p(_ptr(logvar) if want_moments else 0),
                                   C.c_void_p(_ptr(z)), C.c_void_p(_ptr(kl)), C.c_void_p(_stream())))
        return (kl, z, mean, logvar) if want_moments else (kl, z)

    def _decode_latents_eager(self, z, ctx):
        check(lib().rald_ae_decode_latents(self._h, C.c_void_p(_ptr(z)), z.shape[0], C.c_void_p(_ptr(ctx)), C.c_void_p(_stream())))

    def decode_latents(self, z: torch.Tensor, use_graph: Optional[bool] = None) -> torch.Tensor:
        _need_cuda(z, "latents")
        z = _f32c(z)
        B = z.shape[0]
        nbytes = lib().rald_ae_ctx_bytes(self._h, B)
        if use_graph is None:
            use_graph = _graphs_enabled() and B <= GRAPH_MAX_BATCH
        if B > self._dec_batch:                        # the latent-stack workspace grows: captured graphs are stale
            self._graphs.clear()
            self._dec_batch = B
        if not use_graph:
            ctx = torch.empty(nbytes, dtype=torch.uint8, device=z.device)
            self._decode_latents_eager(z, ctx)
            return ctx
        (ctx,) = self._graphs.run(("dec", B), [z], lambda: [torch.empty(nbytes, dtype=torch.uint8, device=z.device)],
                                  lambda ins, outs: self._decode_latents_eager(ins[0], outs[0]))
        return ctx

    def decode_queries(self, ctx: torch.Tensor, queries: torch.Tensor) -> torch.Tensor:
        _need_cuda(queries, "queries")
        queries = _f32c(queries)
        B, Q, _ = queries.shape
        if ctx.dtype != torch.uint8 or not ctx.is_cuda or ctx.numel() != lib().rald_ae_ctx_bytes(self._h, B):
            raise RuntimeError(f"decoder context of {ctx.numel()} bytes does not belong to a batch of {B}: decode_latents(z) and the queries "
                               "must have the same batch size")
        out = torch.empty(B, Q, device=queries.device, dtype=torch.float32)
        check(lib().rald_ae_decode_queries(self._h, C.c_void_p(_ptr(ctx)), C.c_void_p(_ptr(queries)), B, Q, C.c_void_p(_ptr(out)),
                                           C.c_void_p(_stream())))
        return out


# ---- MXFP8 (OCP microscaling: e4m3 elements + one e8m0 scale per 32 K-elements) -----------------------
def op_quantize_mx8(x: torch.Tensor):
    """x [..., K] f32 or bf16 (last dim contiguous) -> (q uint8 [..., K] e4m3 bytes, scales uint8 [..., K/32] e8m0)."""
    _need_cuda(x, "x")
    if x.dtype not in (torch.float32, torch.bfloat16):
        raise TypeError("op_quantize_mx8: float32 or bfloat16 input expected")
    K = x.shape[-1]
    x2 = x.reshape(-1, K)
    if x2.stride(-1) != 1:
        x2 = x2.contiguous()
    rows = x2.shape[0]
    q = torch.empty(rows, K, device=x.device, dtype=torch.uint8)
    s = torch.empty(rows, K // 32, device=x.device, dtype=torch.uint8)
    check(lib().rald_op_quantize_mx8(C.c_void_p(_ptr(x2)), int(x.dtype == torch.bfloat16), x2.stride(0), C.c_void_p(_ptr(q)), K,
                                     C.c_void_p(_ptr(s)), rows, K, C.c_void_p(_stream())))
    return q.reshape(*x.shape), s.reshape(*x.shape[:-1], K // 32)


def op_gemm_mx8(A8: torch.Tensor, sA: torch.Tensor, B8: torch.Tensor, sB: torch.Tensor, bias: Optional[torch.Tensor] = None,
                epilogue: int = 0, C_inout: Optional[torch.Tensor] = None, alpha: float = 1.0) -> torch.Tensor:
    """A8 [batch?,M,K] / B8 [batch?,N,K] uint8 (e4m3) with scales [.., K/32] uint8 (e8m0) -> C = alpha*A.B^T + bias.
    epilogue 0 bf16, 1 f32, 2 f32 accumulate into C_inout."""
    batched = A8.dim() == 3 or B8.dim() == 3
    batch = (A8.shape[0] if A8.dim() == 3 else B8.shape[0]) if batched else 1
    M, K = A8.shape[-2], A8.shape[-1]
    N = B8.shape[-2]
    if B8.shape[-1] != K or sA.shape[-1] != K // 32 or sB.shape[-1] != K // 32 or not (sA.is_contiguous() and sB.is_contiguous()):
        raise ValueError("op_gemm_mx8: operand / scale shapes do not match")
    if epilogue == 2:
        out = C_inout
    else:
        shape = (batch, M, N) if batched else (M, N)
        out = torch.empty(shape, device=A8.device, dtype=torch.bfloat16 if epilogue == 0 else torch.float32)
    check(lib().rald_op_gemm_mx8(C.c_void_p(_ptr(A8)), C.c_void_p(_ptr(sA)), A8.stride(-2), A8.stride(0) if A8.dim() == 3 else 0,
                                 sA.stride(0) if sA.dim() == 3 else 0, C.c_void_p(_ptr(B8)), C.c_void_p(_ptr(sB)), B8.stride(-2),
                                 B8.stride(0) if B8.dim() == 3 else 0, sB.stride(0) if sB.dim() == 3 else 0, C.c_void_p(_ptr(out)),
                                 out.stride(-2), out.stride(0) if out.dim() == 3 else 0, C.c_void_p(_ptr(bias) if bias is not None else 0),
                                 M, N, K, batch, alpha, epilogue, C.c_void_p(_stream())))
    return out


def op_layernorm_mx8(x: torch.Tensor, g: torch.Tensor, b: torch.Tensor, gstride: int = 0, rows_per_group: int = 1,
                     add_one: float = 0.0, eps: float = 1e-5):
    M, D = x.shape
    q = torch.empty(M, D, device=x.device, dtype=torch.uint8)
    s = torch.empty(M, D // 32, device=x.device, dtype=torch.uint8)
    check(lib().rald_op_layernorm_mx8(C.c_void_p(_ptr(x)), C.c_void_p(_ptr(q)), C.c_void_p(_ptr(s)), M, D, C.c_void_p(_ptr(g)),
                                      C.c_void_p(_ptr(b)), gstride, rows_per_group, add_one, eps, C.c_void_p(_stream())))
    return q, s
