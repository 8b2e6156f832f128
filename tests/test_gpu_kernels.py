"""Kernel-level parity on a real MI355X, driven through the C-ABI (rald_op_*).  Each kernel is
compared with a plain fp32 PyTorch statement of the same op.  Exact-integer cases pin the MFMA
fragment / accumulator layouts bit-for-bit (an asymmetric B catches a transposed C-write)."""
import math

import pytest
import torch

from conftest import rel_l2

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def H():
    from rald_amd import _handles
    return _handles


def _ints(shape, lo, hi, seed):
    g = torch.Generator("cpu").manual_seed(seed)
    return torch.randint(lo, hi + 1, shape, generator=g).float()


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (64, 64, 128), (256, 384, 512), (512, 512, 512),
                                   (200, 132, 64), (8192, 512, 512), (1024, 4096, 512), (300, 1000, 1024)])
def test_gemm_exact_integers(H, M, N, K):
    A = _ints((M, K), -3, 3, 1).cuda()
    B = (_ints((N, K), -2, 2, 2) + (torch.arange(N) % 3 == 0).float()[:, None]).cuda()   # asymmetric
    ref = A @ B.t()
    out = H.op_gemm_nt(A.bfloat16(), B.bfloat16(), epilogue=1)
    assert torch.equal(out, ref)
    out16 = H.op_gemm_nt(A.bfloat16(), B.bfloat16(), epilogue=0)
    assert torch.equal(out16.float(), ref.bfloat16().float())


def test_gemm_batched_and_bias_and_alpha(H):
    A = torch.randn(512, 512, device="cuda").bfloat16()            # shared "weights" as the A operand
    Bm = torch.randn(3, 96, 512, device="cuda").bfloat16()
    bias = torch.randn(96, device="cuda")
    out = H.op_gemm_nt(A, Bm, bias=bias, epilogue=1, alpha=0.5)
    ref = 0.5 * torch.einsum("mk,bnk->bmn", A.float(), Bm.float()) + bias
    assert rel_l2(out, ref) < 2e-6


@pytest.mark.parametrize("M", [512, 4096 * 8])
def test_gemm_residual_epilogue(H, M):
    A = torch.randn(M, 2048, device="cuda").bfloat16()
    W = (torch.randn(512, 2048, device="cuda") / 45).bfloat16()
    bias = torch.randn(512, device="cuda")
    x = torch.randn(M, 512, device="cuda")
    ref = x + A.float() @ W.float().t() + bias
    H.op_gemm_nt(A, W, bias=bias, epilogue=2, C_inout=x)
    assert rel_l2(x, ref) < 2e-6


@pytest.mark.parametrize("M", [512, 16384])
def test_gemm_geglu_epilogue(H, M):
    inner = 2048
    A = torch.randn(M, 512, device="cuda").bfloat16()
    W = (torch.randn(2 * inner, 512, device="cuda") / 22).bfloat16()
    bias = torch.randn(2 * inner, device="cuda") * 0.1
    c = torch.arange(inner)
    rowmap = torch.cat([32 * (c // 16) + c % 16, 32 * (c // 16) + 16 + c % 16]).cuda()
    Wp = torch.empty_like(W); Wp[rowmap] = W
    bp = torch.empty_like(bias); bp[rowmap] = bias
    out = H.op_gemm_nt(A, Wp, bias=bp, epilogue=3)
    y = A.float() @ W.float().t() + bias
    ref = y[:, :inner] * torch.nn.functional.gelu(y[:, inner:])
    assert out.shape == (M, inner)
    assert rel_l2(out, ref) < 4e-3          # bf16 output rounding (2^-9 relative per element)


@pytest.mark.parametrize("Bn,M", [(64, 512), (5, 512), (1, 256)])
def test_gemm_softmax64_epilogue_batched(H, Bn, M):
    """The folded cross-attention's first GEMM (dit.hip cond_fold): P = softmax over every aligned group of 64 output columns of A.B^T, in
    exp2 units, with one B per batch entry - the 256x256 engine with whole batch entries dealt to one XCD (Bn = 64) and the 128x128 one -
    against fp32 torch; rows of every group sum to one."""
    g = torch.Generator("cpu").manual_seed(41)
    A = torch.randn(Bn, M, 512, generator=g).cuda().bfloat16()
    Bm = (torch.randn(Bn, 512, 512, generator=g) / 8).cuda().bfloat16()
    out = H.op_gemm_nt(A, Bm, epilogue=4)
    s = torch.einsum("bmk,bnk->bmn", A.float(), Bm.float()).view(Bn, M, 8, 64) * math.log(2.0)
    ref = torch.softmax(s, dim=-1).view(Bn, M, 512)
    assert out.shape == (Bn, M, 512) and out.dtype == torch.bfloat16
    assert rel_l2(out, ref) < 6e-3                                    # bf16 output rounding
    assert float((out.float().view(Bn, M, 8, 64).sum(-1) - 1).abs().max()) < 2e-2


@pytest.mark.parametrize("D", [256, 512, 1024])
def test_layernorm_plain_and_modulated(H, D):
    M = 1000
    x = torch.randn(M, D, device="cuda") * 3 + 1
    g = torch.randn(D, device="cuda"); b = torch.randn(D, device="cuda")
    out = H.op_layernorm(x, g, b)
    ref = torch.nn.functional.layer_norm(x, (D,), g, b)
    assert rel_l2(out, ref) < 4e-3
    # AdaLN: per-group (sample) modulation rows, (1+scale), rows_per_group = 250
    mod = torch.randn(4, 2 * D, device="cuda")
    out = H.op_layernorm(x, mod, mod[:, D:], gstride=2 * D, rows_per_group=250, add_one=1.0)
    grp = torch.arange(M, device="cuda") // 250
    ref = torch.nn.functional.layer_norm(x, (D,)) * (1 + mod[grp, :D]) + mod[grp, D:]
    assert rel_l2(out, ref) < 4e-3


def _attn_ref(q, k, v, heads, scale):
    B, nq, HD = q.shape
    d = HD // heads
    qh = q.float().view(B, nq, heads, d).transpose(1, 2)
    kh = k.float().view(B, -1, heads, d).transpose(1, 2)
    vh = v.float().view(B, -1, heads, d).transpose(1, 2)
    p = (qh @ kh.transpose(-1, -2) * scale).softmax(-1)
    return (p @ vh).transpose(1, 2).reshape(B, nq, HD)


@pytest.mark.parametrize("B,nq,nk,heads", [(2, 512, 512, 8), (1, 128, 64, 8), (3, 512, 64, 8), (1, 512, 1000, 8), (2, 96, 10000, 2)])
def test_attention_d64(H, B, nq, nk, heads):
    HD = heads * 64
    nkp = (nk + 63) // 64 * 64
    g = torch.Generator("cpu").manual_seed(3)
    q = torch.randn(B, nq, HD, generator=g).cuda().bfloat16()
    k = torch.zeros(B, nkp, HD).cuda().bfloat16()
    k[:, :nk] = torch.randn(B, nk, HD, generator=g).cuda().bfloat16()
    v = torch.randn(B, nk, HD, generator=g).cuda().bfloat16()
    k[:, nk:] = 7.0                                   # garbage in the key pad must be masked out
    vt = torch.zeros(B, HD, nkp, device="cuda", dtype=torch.bfloat16)
    vt[:, :, :nk] = v.transpose(1, 2)
    scale = 1.0 / math.sqrt(64)
    out = H.op_attention(q, k, vt, nk, heads, scale)
    ref = _attn_ref(q, k[:, :nk], v, heads, scale)
    assert rel_l2(out, ref) < 6e-3                     # P and O are rounded to bf16


def test_attention_exact_one_hot(H):
    """Scores engineered so softmax is (numerically) one-hot: O must reproduce the selected V
    rows exactly - pins the key<->accumulator-row permutation of the P.V MFMA."""
    B, nq, nk, heads = 1, 64, 96, 1
    q = torch.zeros(B, nq, 64)
    k = torch.full((B, nk, 64), -16.0)
    sel = (torch.arange(nq) * 7 + 3) % nk              # distinct keys (7 is coprime with 96)
    for i in range(nq):
        q[0, i, i] = 16.0
        k[0, sel[i], i] = 16.0                         # query i matches key sel[i]: score 256 vs -256
    v = _ints((B, nk, 64), -8, 8, 5)
    nkp = 128                                          # K rows / Vt columns padded to a multiple of 64
    kp = torch.zeros(B, nkp, 64); kp[:, :nk] = k
    vt = torch.zeros(B, 64, nkp); vt[:, :, :nk] = v.transpose(1, 2)
    out = H.op_attention(q.cuda().bfloat16(), kp.cuda().bfloat16(), vt.cuda().bfloat16(), nk, heads, 1.0)
    assert torch.equal(out.float().cpu()[0], v[0, sel])


def test_attention_key_split_equals_unsplit(H):
    """Few queries x many ragged keys (the AE's 512 latents x 10 000 points): key ranges on separate workgroups + combine
    pass against the single-pass kernel and the torch reference, for several split counts (incl. more splits than tiles)."""
    B, nq, nk, heads = 1, 128, 1000, 2
    HD, nkp = heads * 64, 1024
    g = torch.Generator("cpu").manual_seed(13)
    q = torch.randn(B, nq, HD, generator=g).cuda().bfloat16()
    k = torch.full((B, nkp, HD), 7.0).cuda().bfloat16()
    k[:, :nk] = torch.randn(B, nk, HD, generator=g).cuda().bfloat16()
    v = torch.randn(B, nk, HD, generator=g).cuda().bfloat16()
    vt = torch.zeros(B, HD, nkp, device="cuda", dtype=torch.bfloat16)
    vt[:, :, :nk] = v.transpose(1, 2)
    scale = 1.0 / math.sqrt(64)
    ref = _attn_ref(q, k[:, :nk], v, heads, scale)
    base = H.op_attention(q, k, vt, nk, heads, scale)
    for ks in (2, 5, 16, 40, 0):
        out = H.op_attention_split(q, k, vt, nk, heads, scale, ks)
        assert rel_l2(out, ref) < 6e-3, ks
        assert rel_l2(out, base) < 4e-3, ks                         # P is rounded to bf16 against a different running max


def test_attention_reference_max_moves_lazily(H, monkeypatch):
    """The running max only moves when a tile's scores exceed it by more than 2^8 (attention.hip, RALD_ATTN_LAZY).  Scores that
    climb tile after tile - by less than the threshold, by far more, and falling - must still give the exact softmax, in
    the kernel's own-scale path and in the prescaled-q path (q already times scale*log2 e, what the denoiser feeds it)."""
    B, nq, nk, heads = 1, 64, 512, 1
    g = torch.Generator("cpu").manual_seed(21)
    ramp = torch.tensor([0., 3., 5., 40., 41., 100., 104., 300.]).repeat_interleave(64) + torch.rand(nk, generator=g)
    k = 0.05 * torch.randn(B, nk, 64, generator=g)
    k[0, :, 0] = ramp
    q = 0.05 * torch.randn(B, nq, 64, generator=g)
    q[0, :, 0] = torch.tensor([1.0, 0.5, 0.02, -1.0]).repeat(16)     # steep, moderate, under the threshold, falling
    v = torch.randn(B, nk, 64, generator=g)
    qb, kb, vb = q.cuda().bfloat16(), k.cuda().bfloat16(), v.cuda().bfloat16()
    vt = vb.transpose(1, 2).contiguous()
    for prescaled, scale in ((0, 1.0), (1, math.log(2.0))):
        monkeypatch.setenv("RALD_ATTN_PRESCALED", str(prescaled))
        out = H.op_attention(qb, kb, vt, nk, heads, scale)
        ref = _attn_ref(qb, kb, vb, heads, scale)
        assert torch.isfinite(out.float()).all()
        assert rel_l2(out, ref) < 6e-3, prescaled
    monkeypatch.delenv("RALD_ATTN_PRESCALED")
    out = H.op_attention_vrow(qb, kb, vb, heads, 1.0)
    assert rel_l2(out, _attn_ref(qb, kb, vb, heads, 1.0)) < 6e-3


def test_attention_row_major_v_transposed_lds_read(H):
    """V row-major like K (a column slice of a fused q|k|v buffer), transposed on the LDS read (ds_read_b64_tr_b16):
    exact on a one-hot softmax with integer V (pins the lane / element mapping of the transposed read), and against the
    torch reference on random data, heads interleaved in a [B, N, 3*H*64] buffer."""
    B, nq, nk, heads = 1, 64, 128, 1
    q = torch.zeros(B, nq, 64)
    k = torch.full((B, nk, 64), -16.0)
    sel = (torch.arange(nq) * 7 + 3) % nk
    for i in range(nq):
        q[0, i, i] = 16.0
        k[0, sel[i], i] = 16.0
    v = _ints((B, nk, 64), -8, 8, 5)
    out = H.op_attention_vrow(q.cuda().bfloat16(), k.cuda().bfloat16(), v.cuda().bfloat16(), heads, 1.0)
    assert torch.equal(out.float().cpu()[0], v[0, sel])
    B, n, heads = 3, 512, 8
    HD = heads * 64
    g = torch.Generator("cpu").manual_seed(9)
    qkv = torch.randn(B, n, 3 * HD, generator=g).cuda().bfloat16()
    scale = 1.0 / math.sqrt(64)
    out = H.op_attention_vrow(qkv[:, :, :HD], qkv[:, :, HD:2 * HD], qkv[:, :, 2 * HD:], heads, scale)
    ref = _attn_ref(qkv[:, :, :HD], qkv[:, :, HD:2 * HD], qkv[:, :, 2 * HD:], heads, scale)
    assert rel_l2(out, ref) < 6e-3
    vt = qkv[:, :, 2 * HD:].transpose(1, 2).contiguous()
    old = H.op_attention(qkv[:, :, :HD], qkv[:, :, HD:2 * HD], vt, n, heads, scale)
    assert torch.equal(out, old)                                   # same arithmetic, different operand path


@pytest.mark.parametrize("M,N1,N2", [(4096, 512, 512), (4096, 1536, 512), (1000, 136, 72), (512, 4096, 512), (70, 64, 1024)])
def test_weight_gradient_gemm_contracts_over_rows(H, M, N1, N2):
    """C += A^T.B for row-major A [M,N1], B [M,N2] (gemm_tn.hip: operands staged as they lie, fragments read transposed from LDS, the row
    range split over workgroups, fp32 atomics) + the column sums of A: exact on small integers (pins the transposed-read lane mapping, the
    ragged row tail and the column edges), against fp32 torch on random data, accumulating into a non-zero C, on column slices."""
    A = _ints((M, N1 + 8), -3, 4, 31).cuda().bfloat16()[:, 8:]                # column slices of wider buffers
    B = _ints((M, N2 + 16), -3, 4, 32).cuda().bfloat16()[:, :N2]
    C0 = _ints((N1, N2), -5, 6, 33).cuda()
    cs0 = _ints((N1,), -5, 6, 34).cuda()
    Cm, cs = C0.clone(), cs0.clone()
    H.op_gemm_tn(A, B, Cm, cs)
    assert torch.equal(Cm, C0 + A.float().t() @ B.float())                    # |sums| < 2^24: exact in fp32 whatever the order
    assert torch.equal(cs, cs0 + A.float().sum(0))
    g = torch.Generator("cpu").manual_seed(35)
    A = torch.randn(M, N1, generator=g).cuda().bfloat16(); B = torch.randn(M, N2, generator=g).cuda().bfloat16()
    Cm = torch.zeros(N1, N2, device="cuda")
    H.op_gemm_tn(A, B, Cm)
    assert rel_l2(Cm, A.float().t() @ B.float()) < 2e-6


def test_first_and_last_layer_small_and_large_row_forms():
    """proj_in (:221 + c_in) and final LayerNorm + proj_out + skip/out scaling (:230-232 + :429): both have a small-M and a large-M kernel
    (weights staged in LDS).  Each against fp64 torch, and the large form bit-equal to the small one on the same rows (same summation order)."""
    import ctypes as C
    from rald_amd._lib import lib, check
    L = lib()
    g = torch.Generator("cpu").manual_seed(23)
    M, D, Cc, rpg = 8192 + 40, 512, 32, 512
    ng = (M + rpg - 1) // rpg
    xin = torch.randn(M, Cc, generator=g).cuda(); W = (torch.randn(D, Cc, generator=g) / Cc ** 0.5).cuda()
    coef = (torch.rand(ng, 4, generator=g) + 0.5).cuda()
    p = lambda t: C.c_void_p(t.data_ptr())

    def run_in(rows):
        x = torch.empty(rows, D, device="cuda")
        check(L.rald_op_proj_in(p(xin), p(W), p(x), rows, Cc, D, p(coef), 4, rpg, None))
        return x
    x_big, x_small = run_in(M), run_in(4096)
    cs = coef[:, 0].repeat_interleave(rpg)[:M, None].double()
    assert rel_l2(x_big.double(), cs * (xin.double() @ W.double().t())) < 1e-6
    assert torch.equal(x_big[:4096], x_small)
    x = (torch.randn(M, D, generator=g) * 2 + 0.3).cuda()
    gam, bet = (1 + 0.1 * torch.randn(D, generator=g)).cuda(), (0.1 * torch.randn(D, generator=g)).cuda()
    Wo = (torch.randn(Cc, D, generator=g) / D ** 0.5).cuda()

    def run_out(rows):
        out = torch.empty(rows, Cc, device="cuda")
        check(L.rald_op_final_norm_proj(p(x), p(gam), p(bet), p(Wo), p(xin), p(out), rows, D, Cc, p(coef), 4, rpg, None))
        return out
    o_big, o_small = run_out(M), run_out(4096)
    xn = torch.nn.functional.layer_norm(x.double(), (D,), gam.double(), bet.double())
    ref = coef[:, 1].repeat_interleave(rpg)[:M, None].double() * xin.double() + coef[:, 2].repeat_interleave(rpg)[:M, None].double() * (xn @ Wo.double().t())
    e_big, e_small = rel_l2(o_big.double(), ref), rel_l2(o_small.double(), ref[:4096])
    print(f"final_norm_proj vs fp64: large-M (bf16 hi+lo on MFMA) {e_big:.1e}, one-row fp32 kernel {e_small:.1e}")
    assert e_small < 2e-6 and e_big < 2e-5                     # the split operands keep ~16 mantissa bits: fp32-grade, not bf16 (4e-3)
    assert rel_l2(o_big[:4096], o_small) < 2e-5


@pytest.mark.parametrize("B,nq,nk,heads,shared", [(1, 512, 10000, 8, True), (2, 128, 1000, 1, False), (3, 64, 130, 2, False)])
def test_attention_fp16_shared_key_value_rows(H, B, nq, nk, heads, shared):
    """The folded encoder's attention (ae_encode.hip): fp32 pre-scaled queries, ONE fp16 row per key serving as key and value of every head,
    ragged key count with a zero pad - against torch in fp32, unsplit and with the keys split over workgroups."""
    g = torch.Generator("cpu").manual_seed(17)
    nkp = (nk + 63) // 64 * 64
    kv = torch.zeros(B, nkp, 64)
    kv[:, :nk, :52] = torch.rand(B, nk, 52, generator=g) * 2 - 1
    kv = kv.cuda().half()
    q = (torch.randn(*((nq,) if shared else (B, nq)), heads * 64, generator=g) * 1.5).cuda()
    qb = q[None].expand(B, -1, -1) if shared else q
    kf = kv[:, :nk].float()
    s = torch.einsum("bqhd,bkd->bhqk", qb.half().float().view(B, nq, heads, 64), kf) * math.log(2.0)
    ref = torch.einsum("bhqk,bkd->bqhd", torch.softmax(s, dim=-1), kf).reshape(B, nq, heads * 64)
    for ks in (0, 4, -1):
        out = H.op_attention_f16kv(q, kv, nk, heads, ks, shared_q=shared)
        assert rel_l2(out, ref) < 5e-3, ks                            # O is rounded to bf16, P to fp16


def test_ae_encoder_point_features_and_inverse_std(H):
    """Per point: the 52 Fourier features (hardware sin on revolutions) and 1/std of the point's embedding from the 52x52 factor,
    as the two fp16 rows F and G = rstd.F - against numpy in float64; the pad rows must be zero."""
    import numpy as np
    from rald_amd import synth, weights
    sd = weights.make_state_dict(weights.ae_spec(), seed=0)
    pc = synth.point_cloud(2, 1000, seed=6)
    W = np.concatenate([sd["point_embed.mlp.weight"].double().numpy(), sd["point_embed.mlp.bias"].double().numpy()[:, None]], axis=1)
    Wc = W - W.mean(0, keepdims=True)
    R = np.linalg.cholesky(Wc.T @ Wc / 512 + 1e-18 * np.eye(52)).T
    F, G = H.op_ae_enc_features(pc.cuda(), sd["point_embed.basis"].cuda(), torch.from_numpy(R).float().cuda())
    pts = pc.double().numpy()
    proj = pts @ sd["point_embed.basis"].double().numpy()
    f = np.concatenate([np.sin(proj), np.cos(proj), pts, np.ones(pts.shape[:2] + (1,))], axis=2)
    rstd = 1.0 / np.sqrt(((f @ Wc.T) ** 2).mean(-1) + 1e-5)
    Fr = f.copy(); Fr[..., 51] = 0
    assert F.shape == (2, 1024, 64) and float(F[:, 1000:].abs().max()) == 0 and float(G[:, 1000:].abs().max()) == 0
    assert float(F[:, :, 52:].abs().max()) == 0 and float(G[:, :, 52:].abs().max()) == 0
    assert np.abs(F[:, :1000, :52].double().cpu().numpy() - Fr).max() < 6e-4          # fp16 rounding of values in [-1, 1] + v_sin
    assert rel_l2(G[:, :1000, :52].double().cpu(), torch.from_numpy(f * rstd[..., None])) < 5e-4


@pytest.mark.parametrize("M,K", [(512, 512), (1000, 2048), (32768, 512), (32768, 2048), (640, 256), (300, 576), (25000, 64), (24704, 576)])
def test_gemm_residual_with_fused_layernorm(H, M, K):
    """x += A.W^T + bias and the next (Ada)LayerNorm in one kernel: both outputs against fp32 torch,
    with per-group modulation rows (AdaLN, add_one = 1) - 64-row and 128-row tile configurations."""
    A = torch.randn(M, K, device="cuda").bfloat16()
    W = (torch.randn(512, K, device="cuda") / K ** 0.5).bfloat16()
    bias = torch.randn(512, device="cuda")
    x = torch.randn(M, 512, device="cuda") * 2 + 0.5
    rpg = 250
    ngrp = (M + rpg - 1) // rpg
    mod = torch.randn(ngrp, 1024, device="cuda")
    xref = x + A.float() @ W.float().t() + bias
    grp = torch.arange(M, device="cuda") // rpg
    href = torch.nn.functional.layer_norm(xref, (512,)) * (1 + mod[grp, :512]) + mod[grp, 512:]
    h = H.op_gemm_resid_ln(A, W, bias, x, mod, mod[:, 512:], gstride=1024, rows_per_group=rpg, add_one=1.0)
    assert rel_l2(x, xref) < 2e-6
    assert rel_l2(h, href) < 4e-3


# ---- small-batch fused attention sub-blocks (attn_small.hip) ---------------------------------------------------------------
def _c(t):
    import ctypes as C
    return C.c_void_p(t.data_ptr() if t is not None else 0)


@pytest.mark.parametrize("NL,B", [(512, 1), (512, 3)])
def test_attn_self_proj_partials_vs_torch(NL, B):
    """part[h] = softmax(q_h k_h^T) v_h . Wo[:, 64h:64h+64]^T per head; sum_h part[h] = to_out(attention) without bias."""
    import ctypes as C
    from rald_amd._lib import check, lib
    g = torch.Generator("cpu").manual_seed(3)
    D = 512
    qkv = (torch.randn(B * NL, 3 * D, generator=g) * 0.8).cuda().bfloat16()
    Wo = (torch.randn(D, D, generator=g) / 22).cuda().bfloat16()
    part = torch.full((8, B * NL, D), float("nan"), device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    check(lib().rald_op_attn_self_proj(_c(qkv), 3 * D, _c(Wo), _c(part), NL, 8, B, C.c_void_p(st)))
    f = qkv.float().view(B, NL, 3, 8, 64)
    q, k, v = f[:, :, 0].permute(0, 2, 1, 3), f[:, :, 1].permute(0, 2, 1, 3), f[:, :, 2].permute(0, 2, 1, 3)    # [B,8,NL,64]
    p = torch.softmax(q @ k.transpose(-1, -2) * math.log(2.0), dim=-1)            # q carries log2e: the kernel uses exp2
    o = p @ v                                                                     # [B,8,NL,64]
    ref = torch.stack([o[:, h].reshape(B * NL, 64) @ Wo.float()[:, 64 * h:64 * h + 64].t() for h in range(8)])
    assert torch.isfinite(part).all()
    err = rel_l2(part, ref)
    print("attn_self_proj partials rel_l2", err)
    assert err < 8e-3                                                             # P and O pass through bf16 once each


@pytest.mark.parametrize("B", [1, 4])
def test_xattn_q2_proj_partials_vs_torch(B):
    import ctypes as C
    from rald_amd._lib import check, lib
    g = torch.Generator("cpu").manual_seed(4)
    D, NL, T, L, li = 512, 512, 64, 3, 1                                          # cache of a 3-block model, block 1
    M = B * NL
    hin = torch.randn(M, D, generator=g).cuda().bfloat16()
    Wq = (torch.randn(D, D, generator=g) / 22).cuda().bfloat16()
    Wo = (torch.randn(D, D, generator=g) / 22).cuda().bfloat16()
    Kc = torch.randn(B * T, L * D, generator=g).cuda().bfloat16()                 # [B*T][L*D]
    Vt = torch.randn(B, L * D, T, generator=g).cuda().bfloat16()                  # [B][L*D][T]
    part = torch.full((8, M, D), float("nan"), device="cuda")
    qscale = 0.125 * 1.4426950408889634
    st = torch.cuda.current_stream().cuda_stream
    check(lib().rald_op_xattn_q2_proj(_c(hin), _c(Wq), C.c_void_p(Kc.data_ptr() + li * D * 2), L * D, T * L * D,
                                      C.c_void_p(Vt.data_ptr() + li * D * T * 2), T, L * D * T, _c(Wo), _c(part), M, NL, 8, T, qscale, C.c_void_p(st)))
    q = (hin.float() @ Wq.float().t()).view(B, NL, 8, 64).permute(0, 2, 1, 3)     # [B,8,NL,64]
    k = Kc.float().view(B, T, L, 8, 64)[:, :, li].permute(0, 2, 1, 3)             # [B,8,T,64]
    v = Vt.float().view(B, L, 8, 64, T)[:, li].permute(0, 1, 3, 2)                # [B,8,T,64]
    p = torch.softmax(q @ k.transpose(-1, -2) * 0.125, dim=-1)
    o = p @ v
    ref = torch.stack([o[:, h].reshape(M, 64) @ Wo.float()[:, 64 * h:64 * h + 64].t() for h in range(8)])
    assert torch.isfinite(part).all()
    err = rel_l2(part, ref)
    print("xattn_q2_proj partials rel_l2", err)
    assert err < 1e-2                                                             # q, P and O pass through bf16 once each


def test_reduce_resid_ln_matches_torch():
    import ctypes as C
    from rald_amd._lib import check, lib
    g = torch.Generator("cpu").manual_seed(5)
    M, D, S = 1024, 512, 8
    part = torch.randn(S, M, D, generator=g).cuda()
    bias = torch.randn(D, generator=g).cuda()
    x = torch.randn(M, D, generator=g).cuda()
    mod = torch.randn(2, 2 * D, generator=g).cuda()                               # two samples: per-sample (scale | shift)
    ref_x = x + bias + part.sum(0)
    ln = torch.nn.functional.layer_norm(ref_x, (D,))
    ref_h = torch.cat([ln[:512] * (1 + mod[0, :D]) + mod[0, D:], ln[512:] * (1 + mod[1, :D]) + mod[1, D:]])
    h = torch.empty(M, D, device="cuda", dtype=torch.bfloat16)
    st = torch.cuda.current_stream().cuda_stream
    check(lib().rald_op_reduce_resid_ln(_c(part), S, M * D, _c(bias), _c(x), _c(h), M, _c(mod), C.c_void_p(mod.data_ptr() + D * 4), 2 * D, 512, 1.0, 1e-5,
                                        C.c_void_p(st)))
    assert rel_l2(x, ref_x) < 1e-6
    assert rel_l2(h.float(), ref_h) < 4e-3
