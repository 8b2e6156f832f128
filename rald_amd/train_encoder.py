"""Forward + backward of the radar-spectrum encoder on the HIP kernels (the shipped
configuration trains ``radar_enc`` jointly with the denoiser, ``unfreeze_radar_enc: true``).  Restates
model/models_radar_encoder.py ``Encoder`` (:137-241: conv_in -> 5 levels x 2 ResnetBlocks [+ AttnBlock at the last
level] with Downsample between levels -> mid block/attn/block -> norm_out -> conv_out) and the tokeniser half of
``EDMPrecond.process_radar_cond`` (models_radar_generation.py:363-407) as explicit launches with saved activations, and
their gradients:

  * Conv3d data gradient: the forward implicit-GEMM kernel on dY with flipped/transposed weights
    (``rald_op_conv_pack_weights(dgrad=1)``); Downsample (F.pad(0,1) + k3 s2) spreads dY onto a 2x zero grid first;
  * Conv3d weight gradient: ``dW += dY^T . im2col^T`` per chunk of voxels on the bf16 MFMA GEMM, rows in the parameter's
    own [Cout, Cin*27] order so it accumulates straight into ``param.grad``;
  * GroupNorm(+swish) backward in two streaming passes; AttnBlock (single head over 64 tokens) with thin GEMMs.

Layout: channels-last [B, D, H, W, C]; fp32 trunk, bf16 conv inputs (as the inference path, csrc/radar.hip).
PyTorch owns the buffers only.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Tuple

import torch

from . import train_ops as TO
from ._handles import _stream, op_gemm_nt
from ._lib import check, lib

_p = TO._p
CH_MULT = (1, 1, 2, 2, 4)            # models_radar_encoder.py factories (ch_mult of ae_ch64_mult5_*)


def _st():
    return C.c_void_p(_stream())


def pack_conv(W: torch.Tensor, dgrad: bool = False, pad_to: Optional[int] = None) -> torch.Tensor:
    """W [Cout, Cin, 3, 3, 3] f32 -> packed bf16 for ``conv3d`` ([Cout][27][Cin], or the dgrad form [Cin][27][Cout])."""
    Cout, Cin = W.shape[0], W.shape[1]
    inner = Cout if dgrad else Cin
    pad_to = pad_to or inner
    out = torch.empty((Cin if dgrad else Cout), 27, pad_to, device=W.device, dtype=torch.bfloat16)
    check(lib().rald_op_conv_pack_weights(_p(W.contiguous()), _p(out), Cout, Cin, pad_to, int(dgrad), _st()))
    return out


def conv3d(x16: torch.Tensor, wp: torch.Tensor, bias: torch.Tensor, resid: Optional[torch.Tensor] = None, stride: int = 1, pad: int = 1):
    """x16 bf16 [B, D, H, W, Cin], wp packed [Cout][27][Cin] -> f32 [B, D/s, H/s, W/s, Cout] (+ resid)."""
    B, D, H, W, Cin = x16.shape
    Cout = wp.shape[0]
    out = torch.empty(B, D // stride, H // stride, W // stride, Cout, device=x16.device, dtype=torch.float32)
    check(lib().rald_op_conv3d(_p(x16), _p(wp), _p(bias), _p(resid), _p(out), B, D, H, W, Cin, Cout, stride, pad, _st()))
    return out


def groupnorm(x: torch.Tensor, gamma, beta, swish: bool):
    """x f32 [B, ..., C] -> (bf16 same shape, stats [B, 32, 2] f64 for the backward)."""
    B, Cc = x.shape[0], x.shape[-1]
    S = x.numel() // (B * Cc)
    y = torch.empty(x.shape, device=x.device, dtype=torch.bfloat16)
    stats = torch.empty(B, 32, 2, device=x.device, dtype=torch.float64)
    check(lib().rald_op_groupnorm(_p(x), _p(gamma), _p(beta), _p(y), _p(stats), B, S, Cc, int(swish), _st()))
    return y, stats


def groupnorm_bwd(x, stats, gamma, beta, da, dx, dgamma, dbeta, swish: bool, accumulate: bool):
    B, Cc = x.shape[0], x.shape[-1]
    S = x.numel() // (B * Cc)
    scratch = torch.empty(B, 32, 2, device=x.device, dtype=torch.float64)
    check(lib().rald_op_groupnorm_bwd(_p(x), _p(stats), _p(gamma), _p(beta), _p(da), _p(dx), _p(dgamma), _p(dbeta), _p(scratch), B, S, Cc,
                                      int(swish), int(accumulate), _st()))


def _zero_bias(n, dev):
    return torch.zeros(n, device=dev, dtype=torch.float32)


def conv_dgrad(dy: torch.Tensor, W: torch.Tensor) -> torch.Tensor:
    """Gradient w.r.t. the input of a k3 s1 p1 conv: dy f32 [B, D, H, W, Cout], W the f32 parameter -> f32 [B, D, H, W, Cin]."""
    Cout, Cin = W.shape[0], W.shape[1]
    cpad = -(-Cout // 64) * 64
    B = dy.shape[0]
    M = dy.numel() // Cout
    dy16 = torch.empty(*dy.shape[:-1], cpad, device=dy.device, dtype=torch.bfloat16)
    check(lib().rald_op_pad_channels(_p(dy), _p(dy16), M, Cout, cpad, _st()))
    return conv3d(dy16, pack_conv(W, dgrad=True, pad_to=cpad), _zero_bias(Cin, dy.device), None, 1, 1)


def down_dgrad(dy: torch.Tensor, W: torch.Tensor) -> torch.Tensor:
    """Gradient w.r.t. the input of Downsample (:37-41, F.pad(0,1) + k3 s2 p0): dy [B, OD, OH, OW, C] -> [B, 2OD, 2OH, 2OW, C]."""
    B, OD, OH, OW, Cc = dy.shape
    up = torch.empty(B, 2 * OD, 2 * OH, 2 * OW, Cc, device=dy.device, dtype=torch.bfloat16)
    check(lib().rald_op_zero_insert2(_p(dy), _p(up), B, OD, OH, OW, Cc, _st()))
    return conv3d(up, pack_conv(W, dgrad=True), _zero_bias(W.shape[1], dy.device), None, 1, 2)


def conv_wgrad(dy: torch.Tensor, x16: torch.Tensor, dW: torch.Tensor, dbias: Optional[torch.Tensor], stride: int = 1, pad: int = 1) -> None:
    """dW [Cout, Cin, 3, 3, 3] (f32, accumulated) += sum over voxels of dy (x) patches(x16); dbias += column sums of dy."""
    B, ID, IH, IW, Cin = x16.shape
    Cout = dy.shape[-1]
    M = dy.numel() // Cout
    dy2 = dy.reshape(M, Cout)
    dW2 = dW.view(Cout, Cin * 27)
    nchunk = max(64, min(M, ((1 << 27) // (Cin * 27 * 2)) // 64 * 64))
    for m0 in range(0, M, nchunk):
        n = min(nchunk, M - m0)
        npad = -(-n // 64) * 64                              # the GEMM contracts over the chunk: multiple of 64 (zero rows beyond M)
        col = torch.empty(Cin * 27, npad, device=dy.device, dtype=torch.bfloat16)
        check(lib().rald_op_im2col_t(_p(x16), _p(col), B, ID, IH, IW, Cin, stride, pad, m0, npad, _st()))
        rows = dy2[m0:m0 + n]
        if npad != n:
            rows = torch.cat([rows, torch.zeros(npad - n, Cout, device=dy.device, dtype=dy.dtype)], 0)
        op_gemm_nt(TO.T2(rows), col, epilogue=2, C_inout=dW2)
    if dbias is not None:
        TO.colsum(dy2, dbias)
