"""The algebra behind the streaming query decoder (rald_amd/csrc/ae_decode.hip), checked on the CPU: the weight-only
tables that Ae::finalize builds on the host (rald_op_ae_decode_tables: no GPU involved) are fed to a float64 restatement of
what the kernel computes per query, and the result must equal the oracle's KLAutoEncoder.decode (models_ae.py:417-424) on
the same latents - 'exact in real arithmetic' made testable.  The fp16 / MFMA side of the kernel is covered by the -m gpu
tests against the reference goldens."""
import ctypes as C

import numpy as np
import pytest
import torch

from conftest import rel_l2
from oracle import rald_oracle as O
from rald_amd import synth, weights


def _img_off(row, k):
    return row * 128 + (((k >> 3) ^ (row & 7)) << 4) + (k & 7) * 2


def _slot(f):
    if f < 24:
        return 16 * (f >> 3) + (f & 7)
    if f < 48:
        e = f - 24
        return 16 * (e >> 3) + 8 + (e & 7)
    return 48 + (f - 48)


def _tables(sd, d):
    from rald_amd import _lib
    L = _lib.lib()
    f32 = lambda t: np.ascontiguousarray(t.detach().numpy().astype(np.float32))
    wq = f32(sd["decoder_cross_attn.fn.to_q.weight"])
    wkv = f32(sd["decoder_cross_attn.fn.to_kv.weight"])
    wo, bo = sd["decoder_cross_attn.fn.to_out.weight"].double(), sd["decoder_cross_attn.fn.to_out.bias"].double()
    w_out, b_out = sd["to_outputs.weight"].double()[0], sd["to_outputs.bias"].double()[0]
    wfold = (wkv[d:].astype(np.float64).T @ (wo.T @ w_out).numpy()).astype(np.float32)       # Wv^T.Wo^T.w_out (ae.hip, finalize)
    c0 = float(bo @ w_out + b_out)
    ng, nb = f32(sd["decoder_cross_attn.norm.weight"]), f32(sd["decoder_cross_attn.norm.bias"])
    wpe, bpe = f32(sd["point_embed.mlp.weight"]), f32(sd["point_embed.mlp.bias"])
    t2 = np.zeros((d, 64), np.float32)
    limg = np.zeros(64 * 64, np.uint16)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    rc = L.rald_op_ae_decode_tables(d, p(wq), p(np.ascontiguousarray(wkv[:d])), p(ng), p(nb), p(wpe), p(bpe), p(wfold), p(t2), p(limg))
    assert rc == 0
    Lm = np.zeros((64, 64))
    halves = limg.view(np.float16)
    for i in range(64):
        for k in range(64):
            Lm[i, k] = halves[_img_off(i, k) // 2]
    return t2.astype(np.float64), Lm, c0


@pytest.mark.parametrize("dim,M", [(512, 512), (256, 128)])
def test_streaming_decode_fold_equals_reference_decode(dim, M):
    sd = weights.make_state_dict(weights.ae_spec(dim=dim, num_latents=M), seed=0)
    z = synth.normal([2, M, 32], 5)
    q = synth.queries(2, 700, seed=8)
    with torch.no_grad():
        x = O.ae_latent_stack(sd, z, depth=24)
        ref = O.ae_decode_queries(sd, x, q).squeeze(-1)
    t2, Lm, c0 = _tables(sd, dim)
    xc = torch.nn.functional.layer_norm(x, (dim,), sd["decoder_cross_attn.norm_context.weight"], sd["decoder_cross_attn.norm_context.bias"])
    Y = xc.double().numpy() @ t2                                            # [B, M, 64], slot order
    basis = sd["point_embed.basis"].double().numpy()
    out = np.zeros((2, 700))
    for b in range(2):
        pts = q[b].double().numpy()
        proj = pts @ basis
        feat = np.concatenate([np.sin(proj), np.cos(proj), pts], axis=1)   # reference feature order (:133)
        F = np.zeros((700, 64))
        for f in range(51):
            F[:, _slot(f)] = feat[:, f]
        F[:, 51] = 1.0
        var = ((F @ Lm.T) ** 2).sum(1)                                      # |L.[feat;1]|^2 = variance of the point embedding
        rstd = 1.0 / np.sqrt(var + 1e-5)
        S2 = rstd[:, None] * (F[:, :51] @ Y[b][:, :51].T + Y[b][:, 51][None]) + Y[b][:, 53][None]    # log2 domain
        S2 -= S2.max(1, keepdims=True)
        P = np.exp2(S2)
        out[b] = (P * Y[b][:, 63][None]).sum(1) / P.sum(1) + c0
    # the variance factor is stored in fp16 (the kernel's operand precision): 1e-4 on the logits, not 1e-6
    err = rel_l2(torch.from_numpy(out), ref)
    print("fold vs oracle decode", err)
    assert err < 2e-4


def test_variance_factor_reproduces_the_layernorm_statistics():
    """|L.[feat;1]|^2 against the variance of PointEmbed(q) over its 512 outputs, directly."""
    sd = weights.make_state_dict(weights.ae_spec(), seed=0)
    _, Lm, _ = _tables(sd, 512)
    q = synth.queries(1, 300, seed=9)
    qe = O.point_embed(sd, q)[0].double()
    var_ref = qe.var(dim=1, unbiased=False).numpy()
    pts = q[0].double().numpy()
    proj = pts @ sd["point_embed.basis"].double().numpy()
    feat = np.concatenate([np.sin(proj), np.cos(proj), pts], axis=1)
    F = np.zeros((300, 64))
    for f in range(51):
        F[:, _slot(f)] = feat[:, f]
    F[:, 51] = 1.0
    var = ((F @ Lm.T) ** 2).sum(1)
    assert np.abs(var / var_ref - 1).max() < 2e-3                           # fp16 entries of L
