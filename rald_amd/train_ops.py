"""Backward pass of the denoiser's transformer block on the HIP kernels ("backward
kernels for K2-K5").  ``block_forward`` / ``block_backward`` restate what autograd derives for
``BasicTransformerBlock`` (models_radar_generation.py:133-169: AdaLayerNorm -> self-attention -> AdaLayerNorm ->
radar cross-attention -> AdaLayerNorm -> GEGLU feed-forward, residual after each) as explicit launches:
every product is a bf16 MFMA GEMM through ``rald_op_gemm_nt[2]`` (fp32 accumulate; dX = dY.W with the
transposed weight, dW = dY^T.X with transposed activations), everything else a streaming kernel of
``csrc/train_kernels.hip``.  PyTorch only owns the buffers.

Attention backward is two fused launches (csrc/attn_bwd.hip: a query-side and a key-side kernel that recompute the
probabilities from the log-sum-exp, flash-attention style); shapes they do not cover take the unfused form: S = Q.K^T
and dP = dO.V^T per head as batched K = 64 GEMMs, the softmax backward element-wise, the key-side gradients in the
transposed orientation.

Gradient parity against autograd of the CPU oracle: tests/test_train_block.py (one block, whole denoiser);
the loop over blocks, embeddings and loss live in train_dit.py, the radar encoder in train_encoder.py.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Tuple

import torch

from ._handles import _ptr, _stream, op_attention, op_attention_vrow, op_gemm_nt, op_gemm_tn, op_layernorm
from ._lib import check, lib

HEAD = 64


def _p(t):
    return C.c_void_p(_ptr(t) if t is not None else 0)


def cast_bf16(x: torch.Tensor) -> torch.Tensor:
    out = torch.empty(x.shape, device=x.device, dtype=torch.bfloat16)
    check(lib().rald_op_cast_bf16(_p(x), _p(out), x.numel(), C.c_void_p(_stream())))
    return out


def gemm2(A, lda, sA, sA2, B, ldb, sB, sB2, out, ldc, sC, sC2, M, N, K, batch, batch2, epilogue=0, alpha=1.0, bias=None):
    """C = alpha * A.B^T (+bias) with an outer and an inner batch; operands are (tensor-with-offset, ld, strides)."""
    check(lib().rald_op_gemm_nt2(_p(A), lda, sA, sA2, _p(B), ldb, sB, sB2, _p(out), ldc, sC, sC2, _p(bias), M, N, K, batch, batch2, alpha,
                                 epilogue, C.c_void_p(_stream())))
    return out


def transpose(x: torch.Tensor, rows: int, cols: int, ld_in: int, batch: int = 1, stride_in: int = 0, batch2: int = 1, stride_in2: int = 0):
    """[batch][batch2] matrices of `rows` x `cols` inside x (f32 or bf16) -> bf16 [batch, batch2, cols, rows]."""
    out = torch.empty(batch, batch2, cols, rows, device=x.device, dtype=torch.bfloat16)
    check(lib().rald_op_transpose(_p(x), int(x.dtype == torch.bfloat16), ld_in, stride_in, stride_in2, _p(out), rows, batch2 * cols * rows,
                                  cols * rows, rows, cols, batch, batch2, C.c_void_p(_stream())))
    return out


def T2(x: torch.Tensor) -> torch.Tensor:
    """plain [R, C] -> bf16 [C, R]"""
    return transpose(x, x.shape[0], x.shape[1], x.stride(0)).reshape(x.shape[1], x.shape[0])


def ln_mod_bwd(x, dh, scale, gstride, rows_per_group, add_one, dx, dscale, dshift, eps=1e-5, dx_bf16=None):
    """dx += LayerNorm-mod backward of dh (fp32, in place); dx_bf16 (optional, bf16 [rows, 512]) receives the updated dx rounded to bf16."""
    if dx_bf16 is not None:
        check(lib().rald_op_ln_mod_bwd_cast(_p(x), _p(dh), _p(scale), gstride, rows_per_group, add_one, eps, x.shape[0], x.shape[1], _p(dx),
                                            _p(dx_bf16), _p(dscale), _p(dshift), C.c_void_p(_stream())))
        return
    check(lib().rald_op_ln_mod_bwd(_p(x), _p(dh), _p(scale), gstride, rows_per_group, add_one, eps, x.shape[0], x.shape[1], _p(dx), _p(dscale),
                                   _p(dshift), C.c_void_p(_stream())))


def geglu_fwd(u: torch.Tensor) -> torch.Tensor:
    M, two_i = u.shape
    hid = torch.empty(M, two_i // 2, device=u.device, dtype=torch.bfloat16)
    check(lib().rald_op_geglu_fwd(_p(u), _p(hid), M, two_i // 2, C.c_void_p(_stream())))
    return hid


def geglu_bwd(u: torch.Tensor, dhid: torch.Tensor) -> torch.Tensor:
    du = torch.empty_like(u)
    check(lib().rald_op_geglu_bwd(_p(u), _p(dhid), _p(du), u.shape[0], u.shape[1] // 2, C.c_void_p(_stream())))
    return du


def lin_wgrad(dy: torch.Tensor, x_in: torch.Tensor, dW: torch.Tensor, dbias: torch.Tensor = None) -> None:
    """dW [N1, N2] f32 += dy^T . x_in for dy [M, N1], x_in [M, N2] (bf16, or f32: cast first); dbias [N1] += column sums of dy."""
    if dy.dtype != torch.bfloat16:
        dy = cast_bf16(dy)
    if x_in.dtype != torch.bfloat16:
        x_in = cast_bf16(x_in)
    # large gradients: the row ranges meet in a workspace and are summed in order (no atomics: reproducible, and 59 -> 38 us for ff.net.2 at
    # 4 096 rows); a 512 x 512 gradient has too few ranges for the extra launch to pay (20.9 vs 24.0 us)
    op_gemm_tn(dy, x_in, dW, dbias, atomics=dy.shape[1] * x_in.shape[1] < 512 * 1024)


def colsum(x: torch.Tensor, out: torch.Tensor) -> None:
    """out[n] += sum_m x[m, n]"""
    check(lib().rald_op_colsum(_p(x), int(x.dtype == torch.bfloat16), x.stride(0), x.shape[0], x.shape[1], _p(out), C.c_void_p(_stream())))


def attention_backward(q, ldq, k, ldk, v, ldv, O, dO, Bn: int, H: int, nq: int, nk: int, dq, ld_dq, dk, ld_dk, dv, ld_dv):
    """Gradients of O = softmax(q k^T / 8) v per head (models_radar_generation.py:56-74).  q/k/v/dq/dk/dv are bf16
    tensors (possibly column slices of a fused buffer: pass the slice and its row stride); O, dO [Bn*nq, H*64] bf16.
    Two launches of csrc/attn_bwd.hip (nothing score-shaped in memory) when nq % 128 == 0 and nk % 64 == 0 - the denoiser's
    512 latents x 512 / 64 keys; other shapes take the unfused GEMM form below."""
    if nq % 128 or nk % 64:
        return attention_backward_unfused(q, ldq, k, ldk, v, ldv, O, dO, Bn, H, nq, nk, dq, ld_dq, dk, ld_dk, dv, ld_dv)
    scratch = torch.empty(2, Bn * H * nq, device=O.device, dtype=torch.float32)
    check(lib().rald_op_attention_bwd(_p(q), ldq, nq * ldq, _p(k), ldk, nk * ldk, _p(v), ldv, nk * ldv, _p(O), O.stride(0), nq * O.stride(0),
                                      _p(dO), dO.stride(0), nq * dO.stride(0), _p(dq), ld_dq, nq * ld_dq, _p(dk), ld_dk, nk * ld_dk,
                                      _p(dv), ld_dv, nk * ld_dv, _p(scratch[0]), _p(scratch[1]), nq, nk, H, Bn, HEAD ** -0.5, C.c_void_p(_stream())))


def attention_backward_unfused(q, ldq, k, ldk, v, ldv, O, dO, Bn: int, H: int, nq: int, nk: int, dq, ld_dq, dk, ld_dk, dv, ld_dv):
    """The same gradients as batched GEMMs + element-wise passes (any nq, nk): S = Q.K^T and dP = dO.V^T per head as K = 64 GEMMs
    with fp32 results, the softmax backward element-wise, the key side in the transposed orientation."""
    dev, scale = O.device, HEAD ** -0.5
    f32 = lambda *s: torch.empty(*s, device=dev, dtype=torch.float32)
    b16 = lambda *s: torch.empty(*s, device=dev, dtype=torch.bfloat16)
    S, dP = f32(Bn, H, nq, nk), f32(Bn, H, nq, nk)
    gemm2(q, ldq, nq * ldq, HEAD, k, ldk, nk * ldk, HEAD, S, nk, H * nq * nk, nq * nk, nq, nk, HEAD, Bn, H, epilogue=1)
    gemm2(dO, H * HEAD, nq * H * HEAD, HEAD, v, ldv, nk * ldv, HEAD, dP, nk, H * nq * nk, nq * nk, nq, nk, HEAD, Bn, H, epilogue=1)
    lse, delta = f32(Bn, H, nq), f32(Bn, H, nq)
    check(lib().rald_op_row_lse(_p(S), Bn * H * nq, nk, scale, _p(lse), C.c_void_p(_stream())))
    check(lib().rald_op_rowdot_heads(_p(dO), _p(O), Bn * nq, H, nq, _p(delta), C.c_void_p(_stream())))
    dS = b16(Bn, H, nq, nk)
    check(lib().rald_op_attn_bwd_elem(_p(S), _p(dP), _p(lse), _p(delta), Bn * H, nq, nk, nq, 1, scale, 0, _p(None), _p(dS),
                                      C.c_void_p(_stream())))
    kT = transpose(k, nk, HEAD, ldk, Bn, nk * ldk, H, HEAD)                     # [Bn, H, 64, nk]
    gemm2(dS, nk, H * nq * nk, nq * nk, kT, nk, H * HEAD * nk, HEAD * nk, dq, ld_dq, nq * ld_dq, HEAD, nq, HEAD, nk, Bn, H)
    # key side: transposed orientation
    ST, dPT = f32(Bn, H, nk, nq), f32(Bn, H, nk, nq)
    gemm2(k, ldk, nk * ldk, HEAD, q, ldq, nq * ldq, HEAD, ST, nq, H * nk * nq, nk * nq, nk, nq, HEAD, Bn, H, epilogue=1)
    gemm2(v, ldv, nk * ldv, HEAD, dO, H * HEAD, nq * H * HEAD, HEAD, dPT, nq, H * nk * nq, nk * nq, nk, nq, HEAD, Bn, H, epilogue=1)
    PT, dST = b16(Bn, H, nk, nq), b16(Bn, H, nk, nq)
    check(lib().rald_op_attn_bwd_elem(_p(ST), _p(dPT), _p(lse), _p(delta), Bn * H, nk, nq, nq, 1, scale, 1, _p(PT), _p(dST),
                                      C.c_void_p(_stream())))
    qT = transpose(q, nq, HEAD, ldq, Bn, nq * ldq, H, HEAD)                     # [Bn, H, 64, nq]
    dOT = transpose(dO, nq, HEAD, H * HEAD, Bn, nq * H * HEAD, H, HEAD)
    gemm2(dST, nq, H * nk * nq, nk * nq, qT, nq, H * HEAD * nq, HEAD * nq, dk, ld_dk, nk * ld_dk, HEAD, nk, HEAD, nq, Bn, H)
    gemm2(PT, nq, H * nk * nq, nk * nq, dOT, nq, H * HEAD * nq, HEAD * nq, dv, ld_dv, nk * ld_dv, HEAD, nk, HEAD, nq, Bn, H)


def prepare_block_weights(sd: Dict[str, torch.Tensor], prefix: str, device) -> Dict[str, torch.Tensor]:
    """bf16 compute copies (and their transposes for the dX products) of one block's fp32 master weights."""
    g = lambda n: sd[prefix + n].to(device=device, dtype=torch.float32)
    b16 = lambda t: t.to(torch.bfloat16).contiguous()
    W = {"qkv": b16(torch.cat([g("attn1.to_q.weight"), g("attn1.to_k.weight"), g("attn1.to_v.weight")], 0)),
         "o": b16(g("attn1.to_out.0.weight")), "bo": g("attn1.to_out.0.bias").contiguous(),
         "q2": b16(g("attn2.to_q.weight")), "k2": b16(g("attn2.to_k.weight")), "v2": b16(g("attn2.to_v.weight")),
         "o2": b16(g("attn2.to_out.0.weight")), "bo2": g("attn2.to_out.0.bias").contiguous(),
         "w1": b16(g("ff.net.0.proj.weight")), "b1": g("ff.net.0.proj.bias").contiguous(),
         "w2": b16(g("ff.net.2.weight")), "b2": g("ff.net.2.bias").contiguous()}
    for n in ("qkv", "o", "q2", "k2", "v2", "o2", "w1", "w2"):
        W[n + "T"] = T2(W[n])
    return W


def block_forward(W, x: torch.Tensor, mod: torch.Tensor, cond: torch.Tensor, Bn: int, NL: int, H: int = 8):
    """x [Bn*NL, 512] f32 (updated in place), mod [Bn, 3, 1024] f32 = (scale | shift) of norm1..3 (may be a strided
    view of the whole model's table), cond [Bn*T, Cd] bf16.  Returns the saved activations for ``block_backward``."""
    D, M, T = H * HEAD, Bn * NL, cond.shape[0] // Bn
    sv = {"mod": mod, "cond": cond, "Bn": Bn, "NL": NL, "H": H, "T": T}
    ln = lambda j: op_layernorm(x, mod[:, j, :D], mod[:, j, D:], gstride=mod.stride(0), rows_per_group=NL, add_one=1.0)
    sv["x0"] = x.clone()
    sv["h1"] = ln(0)
    qkv = op_gemm_nt(sv["h1"], W["qkv"])                                          # [M, 1536]
    q3 = qkv.view(Bn, NL, 3 * D)
    if NL % 64 == 0:                                                                # V read row-major from the fused buffer (no transposed copy)
        o1 = op_attention_vrow(q3[:, :, :D], q3[:, :, D:2 * D], q3[:, :, 2 * D:], H, HEAD ** -0.5)
    else:
        vt = transpose(qkv[:, 2 * D:], NL, D, 3 * D, Bn, NL * 3 * D).reshape(Bn, D, NL)
        o1 = op_attention(q3[:, :, :D], q3[:, :, D:2 * D], vt, NL, H, HEAD ** -0.5)
    sv["qkv"], sv["o1"] = qkv, o1.reshape(M, D)
    op_gemm_nt(sv["o1"], W["o"], bias=W["bo"], epilogue=2, C_inout=x)
    sv["x1"] = x.clone()
    sv["h2"] = ln(1)
    sv["q2"] = op_gemm_nt(sv["h2"], W["q2"])
    sv["kc"], sv["vc"] = op_gemm_nt(cond, W["k2"]), op_gemm_nt(cond, W["v2"])     # [Bn*T, 512]
    if T % 64 == 0:
        sv["o2"] = op_attention_vrow(sv["q2"].view(Bn, NL, D), sv["kc"].view(Bn, T, D), sv["vc"].view(Bn, T, D), H, HEAD ** -0.5).reshape(M, D)
    else:
        vct = transpose(sv["vc"], T, D, D, Bn, T * D).reshape(Bn, D, T)
        sv["o2"] = op_attention(sv["q2"].view(Bn, NL, D), sv["kc"].view(Bn, T, D), vct, T, H, HEAD ** -0.5).reshape(M, D)
    op_gemm_nt(sv["o2"], W["o2"], bias=W["bo2"], epilogue=2, C_inout=x)
    sv["x2"] = x.clone()
    sv["h3"] = ln(2)
    sv["u"] = op_gemm_nt(sv["h3"], W["w1"], bias=W["b1"])                         # [M, 4096] = [a | g]
    sv["hid"] = geglu_fwd(sv["u"])
    op_gemm_nt(sv["hid"], W["w2"], bias=W["b2"], epilogue=2, C_inout=x)
    return sv


def block_backward(W, sv, dx: torch.Tensor, dmod: torch.Tensor = None, grads: Dict[str, torch.Tensor] = None, dxb: torch.Tensor = None):
    """dx [M, 512] f32 = gradient w.r.t. the block's output; on return it holds the gradient w.r.t. its input.
    ``dmod`` (same shape AND strides as the ``mod`` given to block_forward, accumulated into) and ``grads`` (fp32
    destinations keyed like ``prepare_block_weights``, accumulated into - e.g. views of the flat gradient) are
    allocated when omitted.  ``dxb`` (optional): a bf16 [M, 512] buffer that holds bf16(dx) on entry and on return (every
    LayerNorm backward of the block writes it beside dx, so no separate cast pass runs between the sub-blocks; the caller hands
    the same buffer from block to block).  Returns (grads, dmod, dcond [Bn*T, Cd] f32)."""
    Bn, NL, H, T, mod, cond = sv["Bn"], sv["NL"], sv["H"], sv["T"], sv["mod"], sv["cond"]
    D, M, dev = H * HEAD, Bn * NL, dx.device
    G: Dict[str, torch.Tensor] = grads if grads is not None else {}
    if dmod is None:
        dmod = torch.zeros_like(mod)
    if dmod.stride() != mod.stride():
        raise ValueError("block_backward: dmod must have the strides of mod")
    zeros = lambda *n: torch.zeros(*n, device=dev, dtype=torch.float32)
    ms = mod.stride(0)

    def lin_bwd(dy, x_in, name, bias=None):
        """G[name] += dy^T . x_in (and G[bias] += column sums of dy): one launch of the row-contracting GEMM (csrc/gemm_tn.hip) on the
        operands as they lie - no transposed copies, the bias gradient from the same pass"""
        if name not in G:
            G[name] = zeros(dy.shape[1], x_in.shape[1])
        if bias is not None and bias not in G:
            G[bias] = zeros(dy.shape[1])
        lin_wgrad(dy, x_in, G[name], G[bias] if bias is not None else None)

    if dxb is None:
        dxb = cast_bf16(dx)

    def ada_bwd(j, x_saved, dh):
        ln_mod_bwd(x_saved, dh, mod[:, j, :D], ms, NL, 1.0, dx, dmod[:, j, :D], dmod[:, j, D:], dx_bf16=dxb)

    # ---- feed-forward: x3 = x2 + hid.W2^T + b2 --------------------------------------------------------
    lin_bwd(dxb, sv["hid"], "w2", "b2")
    du = geglu_bwd(sv["u"], op_gemm_nt(dxb, W["w2T"]))                              # [M, 4096]
    lin_bwd(du, sv["h3"], "w1", "b1")
    ada_bwd(2, sv["x2"], op_gemm_nt(du, W["w1T"], epilogue=1))
    # ---- cross-attention: x2 = x1 + o2.Wo2^T + bo2 ---------------------------------------------------
    lin_bwd(dxb, sv["o2"], "o2", "bo2")
    dO2 = op_gemm_nt(dxb, W["o2T"])
    dq2 = torch.empty(M, D, device=dev, dtype=torch.bfloat16)
    dkc, dvc = torch.empty(Bn * T, D, device=dev, dtype=torch.bfloat16), torch.empty(Bn * T, D, device=dev, dtype=torch.bfloat16)
    attention_backward(sv["q2"], D, sv["kc"], D, sv["vc"], D, sv["o2"], dO2, Bn, H, NL, T, dq2, D, dkc, D, dvc, D)
    lin_bwd(dq2, sv["h2"], "q2")
    lin_bwd(dkc, cond, "k2")
    lin_bwd(dvc, cond, "v2")
    dcond = op_gemm_nt(dkc, W["k2T"], epilogue=1)
    op_gemm_nt(dvc, W["v2T"], epilogue=2, C_inout=dcond)
    ada_bwd(1, sv["x1"], op_gemm_nt(dq2, W["q2T"], epilogue=1))
    # ---- self-attention: x1 = x0 + o1.Wo^T + bo ------------------------------------------------------
    lin_bwd(dxb, sv["o1"], "o", "bo")
    dO1 = op_gemm_nt(dxb, W["oT"])
    qkv = sv["qkv"]
    dqkv = torch.empty_like(qkv)
    attention_backward(qkv[:, :D], 3 * D, qkv[:, D:2 * D], 3 * D, qkv[:, 2 * D:], 3 * D, sv["o1"], dO1, Bn, H, NL, NL,
                       dqkv[:, :D], 3 * D, dqkv[:, D:2 * D], 3 * D, dqkv[:, 2 * D:], 3 * D)
    lin_bwd(dqkv, sv["h1"], "qkv")
    ada_bwd(0, sv["x0"], op_gemm_nt(dqkv, W["qkvT"], epilogue=1))
    return G, dmod, dcond
