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


def conv3d(x16: torch.Tensor, wp: torch.Tensor, bias: torch.Tensor, resid: Optional[torch.Tensor] = None, stride: int = 1, pad: int = 1,
           out_bf16: bool = False):
    """x16 bf16 [B, D, H, W, Cin], wp packed [Cout][27][Cin] -> f32 [B, D/s, H/s, W/s, Cout] (+ resid), or the same as bf16 (no resid)."""
    B, D, H, W, Cin = x16.shape
    Cout = wp.shape[0]
    if out_bf16:
        assert resid is None
        out = torch.empty(B, D // stride, H // stride, W // stride, Cout, device=x16.device, dtype=torch.bfloat16)
        check(lib().rald_op_conv3d_bf16(_p(x16), _p(wp), _p(bias), _p(out), B, D, H, W, Cin, Cout, stride, pad, _st()))
        return out
    out = torch.empty(B, D // stride, H // stride, W // stride, Cout, device=x16.device, dtype=torch.float32)
    check(lib().rald_op_conv3d(_p(x16), _p(wp), _p(bias), _p(resid), _p(out), B, D, H, W, Cin, Cout, stride, pad, _st()))
    return out


def groupnorm(x: torch.Tensor, gamma, beta, swish: bool):
    """x f32 [B, ..., C] -> (bf16 same shape, stats [B, 32, 2] f64 for the backward)."""
    B, Cc = x.shape[0], x.shape[-1]
    S = x.numel() // (B * Cc)
    y = torch.empty(x.shape, device=x.device, dtype=torch.bfloat16)
    buf = torch.empty(B * 64 * (1 + (S + 511) // 512), device=x.device, dtype=torch.float64)   # stats + per-block partials
    check(lib().rald_op_groupnorm(_p(x), _p(gamma), _p(beta), _p(y), _p(buf), B, S, Cc, int(swish), _st()))
    return y, buf[:B * 64].view(B, 32, 2)


def groupnorm_apply(x: torch.Tensor, stats: torch.Tensor, gamma, beta, swish: bool) -> torch.Tensor:
    """The normalisation alone from the statistics ``groupnorm`` returned (the backward pass re-creates the activations it did not keep)."""
    B, Cc = x.shape[0], x.shape[-1]
    S = x.numel() // (B * Cc)
    y = torch.empty(x.shape, device=x.device, dtype=torch.bfloat16)
    check(lib().rald_op_groupnorm_apply(_p(x), _p(stats), _p(gamma), _p(beta), _p(y), B, S, Cc, int(swish), _st()))
    return y


def groupnorm_bwd(x, stats, gamma, beta, da, dx, dgamma, dbeta, swish: bool, accumulate: bool, dx_bf16: Optional[torch.Tensor] = None):
    """dx (fp32, may be None when only ``dx_bf16`` is wanted) written or accumulated; dx_bf16 (optional) = the resulting dx rounded to bf16."""
    B, Cc = x.shape[0], x.shape[-1]
    S = x.numel() // (B * Cc)
    scratch = torch.empty((lib().rald_op_groupnorm_bwd_scratch_bytes(B, S, Cc) + 7) // 8, device=x.device, dtype=torch.float64)
    if dx_bf16 is None:
        assert da.dtype == torch.float32
        check(lib().rald_op_groupnorm_bwd(_p(x), _p(stats), _p(gamma), _p(beta), _p(da), _p(dx), _p(dgamma), _p(dbeta), _p(scratch), B, S, Cc,
                                          int(swish), int(accumulate), _st()))
    else:
        check(lib().rald_op_groupnorm_bwd_cast(_p(x), _p(stats), _p(gamma), _p(beta), _p(da), int(da.dtype == torch.bfloat16), _p(dx), _p(dx_bf16),
                                               _p(dgamma), _p(dbeta), _p(scratch), B, S, Cc, int(swish), int(accumulate), _st()))


def _zero_bias(n, dev):
    return torch.zeros(n, device=dev, dtype=torch.float32)


def conv_dgrad(dy: torch.Tensor, W: torch.Tensor, out_bf16: bool = False) -> torch.Tensor:
    """Gradient w.r.t. the input of a k3 s1 p1 conv: dy f32 or bf16 [B, D, H, W, Cout], W the f32 parameter -> f32 (or bf16) [B, D, H, W, Cin]."""
    Cout, Cin = W.shape[0], W.shape[1]
    cpad = -(-Cout // 64) * 64
    B = dy.shape[0]
    M = dy.numel() // Cout
    if dy.dtype == torch.bfloat16 and cpad == Cout:
        dy16 = dy.contiguous()                               # already what the convolution reads (the caller's one bf16 copy of dy)
    else:
        dy16 = torch.empty(*dy.shape[:-1], cpad, device=dy.device, dtype=torch.bfloat16)
        check(lib().rald_op_pad_channels(_p(dy.float() if dy.dtype != torch.float32 else dy), _p(dy16), M, Cout, cpad, _st()))
    return conv3d(dy16, pack_conv(W, dgrad=True, pad_to=cpad), _zero_bias(Cin, dy.device), None, 1, 1, out_bf16=out_bf16)


def down_dgrad(dy: torch.Tensor, W: torch.Tensor) -> torch.Tensor:
    """Gradient w.r.t. the input of Downsample (:37-41, F.pad(0,1) + k3 s2 p0): dy [B, OD, OH, OW, C] -> [B, 2OD, 2OH, 2OW, C]."""
    B, OD, OH, OW, Cc = dy.shape
    up = torch.empty(B, 2 * OD, 2 * OH, 2 * OW, Cc, device=dy.device, dtype=torch.bfloat16)
    check(lib().rald_op_zero_insert2(_p(dy), _p(up), B, OD, OH, OW, Cc, _st()))
    return conv3d(up, pack_conv(W, dgrad=True), _zero_bias(W.shape[1], dy.device), None, 1, 2)


def conv_wgrad(dy: torch.Tensor, x16: torch.Tensor, dW: torch.Tensor, dbias: Optional[torch.Tensor], stride: int = 1, pad: int = 1) -> None:
    """dW [Cout, Cin, 3, 3, 3] (f32, accumulated) += sum over voxels of dy (x) patches(x16); dbias += column sums of dy.
    One launch of the row-contracting GEMM whose second operand is the VIRTUAL patch matrix (csrc/gemm_tn.hip, CONV form): the
    [voxels x 27 Cin] patches are gathered from the channels-last input by the LDS-DMA itself (padding from a zero line), so no
    im2col buffer (7 GB written per full-resolution conv at B = 8 before), no transposed dy, no split-K partials to sum."""
    B, ID, IH, IW, Cin = x16.shape
    Cout = dy.shape[-1]
    dy16 = dy.reshape(-1, Cout)
    if dy16.dtype != torch.bfloat16:
        dy16 = TO.cast_bf16(dy16.contiguous())
    assert x16.dtype == torch.bfloat16 and x16.is_contiguous() and dW.is_contiguous() and dW.dtype == torch.float32
    assert dy16.shape[0] == B * (ID // stride) * (IH // stride) * (IW // stride)
    # the voxel ranges meet in a workspace and are summed in order by a second launch (no atomics: bit-reproducible, and 4-13 x faster below
    # full resolution - csrc/gemm_tn.hip)
    nbytes = lib().rald_op_conv3d_wgrad_workspace_bytes(B, ID, IH, IW, Cin, Cout, stride, pad)
    ws = torch.empty(max(nbytes, 16), device=dy16.device, dtype=torch.uint8)
    check(lib().rald_op_conv3d_wgrad_ws(_p(dy16), _p(x16), _p(dW), _p(dbias), B, ID, IH, IW, Cin, Cout, stride, pad, _p(ws), nbytes, _st()))


def _g(p: torch.nn.Parameter) -> torch.Tensor:
    if p.grad is None:
        p.grad = torch.zeros_like(p.data)
    return p.grad


def _sgemm(A, B, out, trans_a=False, trans_b=False):
    M = A.shape[1] if trans_a else A.shape[0]
    K = A.shape[0] if trans_a else A.shape[1]
    N = B.shape[1] if trans_b else B.shape[0]
    check(lib().rald_op_sgemm_acc(_p(A), A.stride(0), int(trans_a), _p(B), B.stride(0), int(trans_b), _p(out), out.stride(0), M, N, K, 1.0, _st()))
    return out


class EncoderTrainer:
    """Forward with saved activations + backward of ``Encoder`` (ch 64, ch_mult (1,1,2,2,4), 2 res blocks, attention at the
    last level, z channels 16) and of the tokeniser (radar_token_project + r/a/e embeddings).  ``named_params``: the
    parameters by their EDMPrecond state-dict names (``radar_enc.*``, ``radar_token_project.*``, ``radar_{r,a,e}_emb.weight``);
    gradients are accumulated into ``param.grad``."""

    def __init__(self, named_params: Dict[str, torch.nn.Parameter], ch: int = 64, prefix: str = "radar_enc."):
        self.P, self.ch, self.pre = named_params, ch, prefix
        self.dev = next(iter(named_params.values())).device
        if self.dev.type != "cuda":
            raise RuntimeError("EncoderTrainer runs on the HIP device only (no CPU fallback)")
        self.saved: List = []

    def p(self, name: str) -> torch.nn.Parameter:
        return self.P[self.pre + name]

    def w(self, name: str) -> torch.Tensor:
        return self.p(name).data

    # ---- ResnetBlock :46-100 ---------------------------------------------------------------------------------------
    def _res_fwd(self, x, name, cin, cout):
        h1, st1 = groupnorm(x, self.w(name + ".norm1.weight"), self.w(name + ".norm1.bias"), True)
        t1 = conv3d(h1, pack_conv(self.w(name + ".conv1.weight")), self.w(name + ".conv1.bias"))
        h2, st2 = groupnorm(t1, self.w(name + ".norm2.weight"), self.w(name + ".norm2.bias"), True)
        res = x
        if cin != cout:
            x16 = TO.cast_bf16(x)
            res = op_gemm_nt(x16.view(-1, cin), self.w(name + ".nin_shortcut.weight").view(cout, cin).to(torch.bfloat16),
                             bias=self.w(name + ".nin_shortcut.bias"), epilogue=1).view(*x.shape[:-1], cout)
        out = conv3d(h2, pack_conv(self.w(name + ".conv2.weight")), self.w(name + ".conv2.bias"), resid=res)
        # the normalised activations are kept for the weight gradients (bf16: half the size of x / t1 beside them; 3 GB per iteration at
        # B = 8 of the 288 GB) instead of being re-created in the backward pass
        self.saved.append(("res", name, cin, cout, x, st1, t1, st2, h1, h2))
        return out

    def _res_bwd(self, rec, dout, dout16=None):
        """dout: fp32 gradient w.r.t. the block's output, dout16 its bf16 copy when the producer already made one.  The convolution
        gradients read bf16: every gradient tensor is rounded ONCE (by the GroupNorm backward that produces it, in the same pass) and
        that copy feeds both the weight- and the data-gradient convolution.  Returns (dx fp32, dx bf16)."""
        _, name, cin, cout, x, st1, t1, st2, h1, h2 = rec
        P = lambda n: self.p(name + n)
        if dout16 is None:
            dout16 = TO.cast_bf16(dout)
        conv_wgrad(dout16, h2, _g(P(".conv2.weight")), _g(P(".conv2.bias")))
        del h2
        dh2 = conv_dgrad(dout16, P(".conv2.weight").data, out_bf16=True)     # read only by the GroupNorm backward
        dt1 = torch.empty(t1.shape, device=t1.device, dtype=torch.bfloat16)                 # consumed by the two convolution gradients only
        groupnorm_bwd(t1, st2, P(".norm2.weight").data, P(".norm2.bias").data, dh2, None, _g(P(".norm2.weight")), _g(P(".norm2.bias")), True, False,
                      dx_bf16=dt1)
        del dh2
        conv_wgrad(dt1, h1, _g(P(".conv1.weight")), _g(P(".conv1.bias")))
        del h1
        dh1 = conv_dgrad(dt1, P(".conv1.weight").data, out_bf16=True)
        if cin == cout:
            dx = dout
        else:
            d16 = dout16.view(-1, cout)
            W16 = P(".nin_shortcut.weight").data.view(cout, cin).to(torch.bfloat16)
            dx = op_gemm_nt(d16, TO.T2(W16), epilogue=1).view(*x.shape)
            TO.lin_wgrad(d16, TO.cast_bf16(x).view(-1, cin), _g(P(".nin_shortcut.weight")).view(cout, cin), _g(P(".nin_shortcut.bias")))
        dx16 = torch.empty(x.shape, device=x.device, dtype=torch.bfloat16)
        groupnorm_bwd(x, st1, P(".norm1.weight").data, P(".norm1.bias").data, dh1, dx, _g(P(".norm1.weight")), _g(P(".norm1.bias")), True, True,
                      dx_bf16=dx16)
        return dx, dx16

    # ---- AttnBlock :102-135 (single head, S tokens, scale C^-1/2) -----------------------------------------------------
    def _attn_fwd(self, x, name):
        B, Cc = x.shape[0], x.shape[-1]
        S = x.numel() // (B * Cc)
        W16 = lambda n: self.w(name + n).view(Cc, Cc).to(torch.bfloat16)
        n16, st = groupnorm(x, self.w(name + ".norm.weight"), self.w(name + ".norm.bias"), False)
        n2 = n16.view(B * S, Cc)
        q = op_gemm_nt(n2, W16(".q.weight"), bias=self.w(name + ".q.bias"))
        k = op_gemm_nt(n2, W16(".k.weight"), bias=self.w(name + ".k.bias"))
        v = op_gemm_nt(n2, W16(".v.weight"), bias=self.w(name + ".v.bias"))
        scale = float(Cc) ** -0.5
        Sm = torch.empty(B, S, S, device=x.device, dtype=torch.float32)
        TO.gemm2(q, Cc, S * Cc, 0, k, Cc, S * Cc, 0, Sm, S, S * S, 0, S, S, Cc, B, 1, epilogue=1)
        lse = torch.empty(B, S, device=x.device, dtype=torch.float32)
        check(lib().rald_op_row_lse(_p(Sm), B * S, S, scale, _p(lse), _st()))
        Pm = torch.empty(B, S, S, device=x.device, dtype=torch.bfloat16)
        Ssc = Sm * scale                                                # [B,64,64]: tiny
        check(lib().rald_op_softmax_rows(_p(Ssc), S, _p(Pm), S, B * S, S, _st()))
        vT = TO.transpose(v, S, Cc, Cc, B, S * Cc).view(B, Cc, S)
        o = torch.empty(B * S, Cc, device=x.device, dtype=torch.bfloat16)
        TO.gemm2(Pm, S, S * S, 0, vT, S, Cc * S, 0, o, Cc, S * Cc, 0, S, Cc, S, B, 1)
        out = x.clone()
        op_gemm_nt(o, W16(".proj_out.weight"), bias=self.w(name + ".proj_out.bias"), epilogue=2, C_inout=out.view(B * S, Cc))
        self.saved.append(("attn", name, x, st, q, k, v, Sm, lse, Pm, o))
        return out

    def _attn_bwd(self, rec, dxo):
        _, name, x, st, q, k, v, Sm, lse, Pm, o = rec
        B, Cc = x.shape[0], x.shape[-1]
        S = x.numel() // (B * Cc)
        P = lambda n: self.p(name + n)
        W16 = lambda n: P(n).data.view(Cc, Cc).to(torch.bfloat16)
        scale = float(Cc) ** -0.5
        d2 = dxo.view(B * S, Cc)
        n16 = groupnorm_apply(x, st, P(".norm.weight").data, P(".norm.bias").data, False)
        n2 = n16.view(B * S, Cc)
        do = op_gemm_nt(TO.cast_bf16(d2), TO.T2(W16(".proj_out.weight")))                       # [B*S, C] bf16
        TO.lin_wgrad(d2, o, _g(P(".proj_out.weight")).view(Cc, Cc), _g(P(".proj_out.bias")))
        dP = torch.empty(B, S, S, device=x.device, dtype=torch.float32)
        TO.gemm2(do, Cc, S * Cc, 0, v, Cc, S * Cc, 0, dP, S, S * S, 0, S, S, Cc, B, 1, epilogue=1)
        delta = torch.empty(B, S, device=x.device, dtype=torch.float32)
        check(lib().rald_op_rowdot(_p(do), _p(o), B * S, Cc, _p(delta), _st()))
        dS = torch.empty(B, S, S, device=x.device, dtype=torch.bfloat16)
        check(lib().rald_op_attn_bwd_elem(_p(Sm), _p(dP), _p(lse), _p(delta), B, S, S, S, 1, scale, 0, _p(None), _p(dS), _st()))
        tb = lambda t, r, c: TO.transpose(t, r, c, c, B, r * c).view(B, c, r)                  # per-sample transpose
        dq = torch.empty(B * S, Cc, device=x.device, dtype=torch.bfloat16)
        dk, dv = torch.empty_like(dq), torch.empty_like(dq)
        TO.gemm2(dS, S, S * S, 0, tb(k, S, Cc), S, Cc * S, 0, dq, Cc, S * Cc, 0, S, Cc, S, B, 1)              # dS . k
        TO.gemm2(tb(dS, S, S), S, S * S, 0, tb(q, S, Cc), S, Cc * S, 0, dk, Cc, S * Cc, 0, S, Cc, S, B, 1)    # dS^T . q
        TO.gemm2(tb(Pm, S, S), S, S * S, 0, tb(do, S, Cc), S, Cc * S, 0, dv, Cc, S * Cc, 0, S, Cc, S, B, 1)   # P^T . do
        dn = op_gemm_nt(dq, TO.T2(W16(".q.weight")), epilogue=1)
        op_gemm_nt(dk, TO.T2(W16(".k.weight")), epilogue=2, C_inout=dn)
        op_gemm_nt(dv, TO.T2(W16(".v.weight")), epilogue=2, C_inout=dn)
        for g, nm in ((dq, ".q"), (dk, ".k"), (dv, ".v")):
            TO.lin_wgrad(g, n2, _g(P(nm + ".weight")).view(Cc, Cc), _g(P(nm + ".bias")))
        dx = dxo                                                                         # residual path
        groupnorm_bwd(x, st, P(".norm.weight").data, P(".norm.bias").data, dn.view(*x.shape), dx, _g(P(".norm.weight")), _g(P(".norm.bias")),
                      False, True)
        return dx

    # ---- Encoder.forward :216-241 + tokeniser (models_radar_generation.py:363-407) -------------------------------------
    def forward(self, cube: torch.Tensor) -> torch.Tensor:
        """cube [B, R, A, E, cube_ch] f32 -> condition tokens [B, R/16*A/16*E/16, 512] f32 (activations saved for backward)."""
        self.saved = []
        B, R, A, E, cch = cube.shape
        ch = self.ch
        cube = cube.contiguous()
        x = torch.empty(B, R, A, E, ch, device=self.dev, dtype=torch.float32)
        check(lib().rald_op_conv_in(_p(cube), cch, 1, _p(self.w("conv_in.weight")), _p(self.w("conv_in.bias")), _p(x), B, R, A, E, ch, _st()))
        self.saved.append(("conv_in", cube))
        cin = ch
        for l in range(5):
            cout = ch * CH_MULT[l]
            for b in range(2):
                x = self._res_fwd(x, f"down.{l}.block.{b}", cin, cout)
                cin = cout
                if l == 4:
                    x = self._attn_fwd(x, f"down.{l}.attn.{b}")
            if l != 4:
                x16 = TO.cast_bf16(x)
                name = f"down.{l}.downsample.conv"
                x = conv3d(x16, pack_conv(self.w(name + ".weight")), self.w(name + ".bias"), stride=2, pad=0)
                self.saved.append(("down", name, x16))
        x = self._res_fwd(x, "mid.block_1", cin, cin)
        x = self._attn_fwd(x, "mid.attn_1")
        x = self._res_fwd(x, "mid.block_2", cin, cin)
        h, st = groupnorm(x, self.w("norm_out.weight"), self.w("norm_out.bias"), True)
        z = conv3d(h, pack_conv(self.w("conv_out.weight")), self.w("conv_out.bias"))           # [B, 8, 4, 2, 16]
        self.saved.append(("out", x, st))
        # tokeniser: Linear(16 -> 512) + r/a/e embeddings, tokens r-major then a then e
        r_e, a_e, e_e = self.P["radar_r_emb.weight"].data, self.P["radar_a_emb.weight"].data, self.P["radar_e_emb.weight"].data
        emb = (r_e[:, None, None, :] + a_e[None, :, None, :] + e_e[None, None, :, :]).reshape(-1, r_e.shape[1])     # [64, 512]
        tok = (emb + self.P["radar_token_project.bias"].data)[None].repeat(B, 1, 1).contiguous()
        z2 = z.view(-1, z.shape[-1])
        _sgemm(z2, self.P["radar_token_project.weight"].data, tok.view(-1, tok.shape[-1]))
        self.saved.append(("tok", z2, (r_e.shape[0], a_e.shape[0], e_e.shape[0])))
        return tok

    def backward(self, dtok: torch.Tensor) -> None:
        """dtok [B, T, 512] f32: gradient w.r.t. the tokens ``forward`` returned.  Accumulates every parameter gradient."""
        P = self.P
        _, z2, (nr, na, ne) = self.saved.pop()
        B = dtok.shape[0]
        d2 = dtok.reshape(-1, dtok.shape[-1]).contiguous()
        _sgemm(d2, z2, _g(P["radar_token_project.weight"]), trans_a=True, trans_b=True)          # [512, 16] = dtok^T . z
        TO.colsum(d2, _g(P["radar_token_project.bias"]))
        # embedding rows: one-hot selection matrices [B*T, n] (token t = (r*na + a)*ne + e)
        t = torch.arange(nr * na * ne, device=self.dev)
        for idx, n, key in ((t // (na * ne), nr, "radar_r_emb.weight"), ((t // ne) % na, na, "radar_a_emb.weight"), (t % ne, ne, "radar_e_emb.weight")):
            sel = torch.nn.functional.one_hot(idx, n).to(torch.float32).repeat(B, 1).contiguous()
            _sgemm(sel, d2, _g(P[key]), trans_a=True, trans_b=True)
        dz = torch.zeros(z2.shape, device=self.dev, dtype=torch.float32)
        _sgemm(d2, P["radar_token_project.weight"].data, dz, trans_b=True)                       # dtok . Wp
        _, x, st = self.saved.pop()
        dz5 = dz.view(*x.shape[:-1], dz.shape[-1])
        h = groupnorm_apply(x, st, self.w("norm_out.weight"), self.w("norm_out.bias"), True)
        conv_wgrad(dz5, h, _g(self.p("conv_out.weight")), _g(self.p("conv_out.bias")))
        dh = conv_dgrad(dz5, self.w("conv_out.weight"), out_bf16=True)
        dx, dx16 = torch.empty_like(x), torch.empty(x.shape, device=x.device, dtype=torch.bfloat16)
        groupnorm_bwd(x, st, self.w("norm_out.weight"), self.w("norm_out.bias"), dh, dx, _g(self.p("norm_out.weight")), _g(self.p("norm_out.bias")),
                      True, False, dx_bf16=dx16)
        while self.saved:
            rec = self.saved.pop()
            kind = rec[0]
            if kind == "res":
                dx, dx16 = self._res_bwd(rec, dx, dx16)
            elif kind == "attn":
                dx, dx16 = self._attn_bwd(rec, dx), None
            elif kind == "down":
                _, name, x16 = rec
                conv_wgrad(dx16 if dx16 is not None else dx, x16, _g(self.p(name + ".weight")), _g(self.p(name + ".bias")), stride=2, pad=0)
                dx, dx16 = down_dgrad(dx, self.w(name + ".weight")), None
            elif kind == "conv_in":
                cube = rec[1]
                Bc, R, A, E, cch = cube.shape
                # dW [ch][1][27] = dy^T . patches (27-neighbourhoods of the one input channel as bf16 rows of 32), dbias from the same launch
                pat = torch.empty(Bc * R * A * E, 32, device=cube.device, dtype=torch.bfloat16)
                check(lib().rald_op_patches27(_p(cube), cch, _p(pat), Bc, R, A, E, _st()))
                dw32 = torch.zeros(self.ch, 32, device=cube.device, dtype=torch.float32)
                TO.lin_wgrad(dx.view(-1, self.ch), pat, dw32, _g(self.p("conv_in.bias")))
                _g(self.p("conv_in.weight")).view(self.ch, 27).add_(dw32[:, :27])
