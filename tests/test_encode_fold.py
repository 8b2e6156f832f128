"""The algebra behind the folded encoder (rald_amd/csrc/ae_encode.hip), checked on the CPU: the weight-only tables that
Ae::finalize builds on the host (rald_op_ae_encode_tables: no GPU involved) are fed to a float64 restatement of what the
device path computes - per point one row of 52 Fourier features, two head-dim-64 attentions with key = value = that row -
and the result must equal the oracle's KLAutoEncoder.encode (models_ae.py:351-399) on the same cloud: 'exact in real
arithmetic' made testable.  The fp16 / MFMA side is covered by the -m gpu tests against the reference goldens."""
import ctypes as C

import numpy as np
import pytest
import torch

from conftest import rel_l2
from oracle import rald_oracle as O
from rald_amd import synth, weights

LOG2E = 1.4426950408889634


def _tables(sd, d, M, heads, mix):
    from rald_amd import _lib
    L = _lib.lib()
    f32 = lambda k: np.ascontiguousarray(sd[k].detach().numpy().astype(np.float32)) if k in sd else None
    names = ["point_embed.mlp.weight", "point_embed.mlp.bias", "d_latents.weight", "mix_attn_layer.norm.weight", "mix_attn_layer.norm.bias",
             "mix_attn_layer.fn.to_q.weight", "mix_attn_layer.fn.to_kv.weight", "mix_attn_layer.fn.to_out.weight", "mix_attn_layer.fn.to_out.bias",
             "s_latents.weight" if mix else "latents.weight", "query_proj.weight", "query_proj.bias",
             "cross_attend_blocks.0.norm_context.weight", "cross_attend_blocks.0.norm_context.bias", "cross_attend_blocks.0.fn.to_q.weight",
             "cross_attend_blocks.0.fn.to_kv.weight", "cross_attend_blocks.0.fn.to_out.weight", "cross_attend_blocks.0.fn.to_out.bias"]
    ins = [f32(n) for n in names]
    I = heads * 64
    outs = [np.zeros(s, np.float32) for s in ((52, 52), (M, I), (d, I), (M, d), (d, 64), (d, 64), (d,))]
    pin = (C.c_void_p * 18)(*[a.ctypes.data if a is not None else None for a in ins])
    pout = (C.c_void_p * 7)(*[a.ctypes.data for a in outs])
    assert L.rald_op_ae_encode_tables(d, M, heads, int(mix), pin, pout) == 0
    return [a.astype(np.float64) for a in outs]


def _softmax2(S):
    S = S - S.max(-1, keepdims=True)
    P = np.exp2(S)
    return P / P.sum(-1, keepdims=True)


@pytest.mark.parametrize("dim,M,mix", [(512, 512, True), (256, 128, True), (256, 128, False)])
def test_folded_encoder_equals_reference_encode(dim, M, mix):
    heads, P = 8, 1500
    spec = weights.ae_spec(dim=dim, num_latents=M) if mix else weights.ae_spec(dim=dim, num_latents=M, query_type="learnable")
    sd = weights.make_state_dict(spec, seed=0)
    pc = synth.point_cloud(2, P, seed=4)
    sdd = {k: v.double() for k, v in sd.items()}
    with torch.no_grad():
        mean_ref, logvar_ref = O.ae_encode_moments(sdd, pc.double())
    Rf, Q1, T4, X0, T1, T3, c3 = _tables(sd, dim, M, heads, mix)
    pts = pc.double().numpy()
    proj = pts @ sd["point_embed.basis"].double().numpy()
    f = np.concatenate([np.sin(proj), np.cos(proj), pts, np.ones(pts.shape[:2] + (1,))], axis=2)           # [B,P,52]
    rstd = 1.0 / np.sqrt(((f @ Rf.T) ** 2).sum(-1) + 1e-5)
    F = f.copy(); F[..., 51] = 0.0
    G = f * rstd[..., None]
    if mix:
        O1 = np.zeros((2, M, heads * 64))
        for h in range(heads):
            Pm = _softmax2(np.einsum("mk,bpk->bmp", Q1[:, 64 * h:64 * h + 52], F))
            O1[:, :, 64 * h:64 * h + 52] = np.einsum("bmp,bpk->bmk", Pm, F)
        x = X0[None] + O1 @ T4.T
    else:
        x = np.broadcast_to(X0[None], (2, M, dim)).copy()
    ng, nb = sdd["cross_attend_blocks.0.norm.weight"].numpy(), sdd["cross_attend_blocks.0.norm.bias"].numpy()
    xn = (x - x.mean(-1, keepdims=True)) / np.sqrt(x.var(-1, keepdims=True) + 1e-5) * ng + nb
    Pm = _softmax2(np.einsum("bmk,bpk->bmp", xn @ T1[:, :52], G))
    x = x + np.einsum("bmp,bpk->bmk", Pm, G) @ T3[:, :52].T + c3
    xt = torch.from_numpy(x)
    with torch.no_grad():
        xt = O.ae_ff(sdd, "cross_attend_blocks.1.", xt) + xt
        mean, logvar = O._lin(sdd, "mean_fc", xt), O._lin(sdd, "logvar_fc", xt)
    e1, e2 = rel_l2(mean, mean_ref), rel_l2(logvar, logvar_ref)
    print(f"folded encoder vs oracle (float64, fp32 tables): mean {e1:.2e}, logvar {e2:.2e}")
    assert e1 < 2e-5 and e2 < 2e-5
