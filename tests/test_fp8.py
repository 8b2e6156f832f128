"""MXFP8 path (BASELINE config #5, "fp8 MFMA QKV/proj"): quantiser and GEMM on
v_mfma_scale_f32_16x16x128_f8f6f4.  CPU: the MX oracle's element rounding against torch's own
float8_e4m3fn conversion and format identities.  GPU: the HIP quantiser bit-exact against the oracle; the
GEMM exact on integer data with power-of-two block scales, and within the MFMA's accumulation error
(1e-4 relative to the output scale) of the oracle product on random data; edge shapes."""
import numpy as np
import pytest
import torch

from rald_amd import synth


@pytest.fixture(autouse=True)
def _inference_mode():
    """The MXFP8 modes belong to the INFERENCE kernels (with gradients enabled EDMPrecond.forward takes the differentiable bf16
    training route, tests/test_gpu_autograd.py)."""
    with torch.no_grad():
        yield


def test_mx_oracle_format_identities():
    from oracle import mx_oracle as MX
    x = synth.normal([64, 512], 300) * torch.logspace(-6, 6, 64)[:, None]
    q, s = MX.quantize_mx8(x)
    assert q.dtype == torch.uint8 and s.shape == (64, 16)
    d = MX.dequantize_mx8(q, s)
    blk = x.reshape(64, 16, 32)
    amax = blk.abs().amax(-1)
    scale = torch.pow(2.0, s.float() - 127)
    assert torch.all(amax <= 448 * scale) and torch.all(amax > 224 * scale)             # smallest power of two that fits
    err = (d - x).reshape(64, 16, 32).abs()
    assert torch.all(err <= 16 * scale[..., None] * (1 + 1e-6))                          # half an ulp of the top binade [256, 448]
    # powers of two and small integers survive exactly
    ints = torch.randint(-8, 9, (16, 128), generator=torch.Generator().manual_seed(5)).float()
    qi, si = MX.quantize_mx8(ints)
    assert torch.equal(MX.dequantize_mx8(qi, si), ints)
    z, sz = MX.quantize_mx8(torch.zeros(2, 64))
    assert int(z.max()) == 0 and int(sz.max()) == 0


@pytest.mark.gpu
def test_hip_quantize_mx8_bit_exact_vs_oracle():
    from oracle import mx_oracle as MX
    from rald_amd import _handles as H
    x = synth.normal([300, 512], 301) * torch.logspace(-5, 5, 300)[:, None]
    x[7] = 0.0
    x[8, 40:72] = 0.0
    for t in (x, x.bfloat16(), synth.normal([3, 5, 64], 302), synth.normal([129, 1024], 303)):
        q, s = H.op_quantize_mx8(t.cuda())
        qo, so = MX.quantize_mx8(t.float())
        assert torch.equal(s.cpu(), so) and torch.equal(q.cpu(), qo)
    g, b = synth.normal([2, 512], 304) * 0.1, synth.normal([2, 512], 305) * 0.1
    xx = synth.normal([2 * 96, 512], 306) * 3
    q, s = H.op_layernorm_mx8(xx.cuda(), g.cuda(), b.cuda(), gstride=512, rows_per_group=96, add_one=1.0)
    ref = torch.nn.functional.layer_norm(xx, (512,)).reshape(2, 96, 512) * (1 + g[:, None]) + b[:, None]
    d = MX.dequantize_mx8(q.cpu(), s.cpu()).reshape(2, 96, 512)
    qo, so = MX.quantize_mx8(ref)
    # LN arithmetic differs in the last fp32 bits from torch's, which can flip an e4m3 rounding: compare values
    assert float((d - MX.dequantize_mx8(qo, so)).abs().max()) <= float(ref.abs().max()) * 2 ** -3
    assert float((d - ref).norm() / ref.norm()) < 4e-2
    assert (s.cpu().reshape(2, 96, 16) == so).float().mean() > 0.999


@pytest.mark.gpu
def test_hip_gemm_mx8_exact_on_integers_and_vs_oracle():
    from oracle import mx_oracle as MX
    from rald_amd import _handles as H
    gen = torch.Generator().manual_seed(11)
    for (M, N, K, batch) in ((512, 512, 512, 1), (256, 768, 128, 1), (200, 260, 256, 1), (64, 512, 1024, 1), (512, 256, 512, 3)):
        shp = lambda r: (batch, r, K) if batch > 1 else (r, K)
        ia = torch.randint(-4, 5, shp(M), generator=gen).float()
        ib = torch.randint(-4, 5, shp(N), generator=gen).float()
        # random power-of-two block scales on top of integer data: exact in fp32
        pa = torch.pow(2.0, torch.randint(-3, 4, (*shp(M)[:-1], K // 32), generator=gen).float())
        pb = torch.pow(2.0, torch.randint(-3, 4, (*shp(N)[:-1], K // 32), generator=gen).float())
        A = (ia.reshape(*pa.shape, 32) * pa[..., None]).reshape(shp(M))
        B = (ib.reshape(*pb.shape, 32) * pb[..., None]).reshape(shp(N))
        qa, sa = H.op_quantize_mx8(A.cuda())
        qb, sb = H.op_quantize_mx8(B.cuda())
        bias = torch.randint(-3, 4, (N,), generator=gen).float()
        out = H.op_gemm_mx8(qa, sa, qb, sb, bias=bias.cuda(), epilogue=1).cpu()
        want = (A.double() @ B.double().transpose(-1, -2) + bias.double()).float()
        assert torch.equal(out, want), (M, N, K, batch)
        acc = torch.ones_like(out).cuda()
        H.op_gemm_mx8(qa, sa, qb, sb, bias=bias.cuda(), epilogue=2, C_inout=acc)
        assert torch.equal(acc.cpu(), want + 1)
    # random data: the MFMA sees exactly the dequantised operands
    A, B = synth.normal([1024, 512], 310) * 2, synth.normal([1536, 512], 311) / 512 ** 0.5
    qa, sa = H.op_quantize_mx8(A.cuda())
    qb, sb = H.op_quantize_mx8(B.cuda())
    out = H.op_gemm_mx8(qa, sa, qb, sb, epilogue=1, alpha=0.125).cpu().double()
    want = MX.gemm_mx8(qa.cpu(), sa.cpu(), qb.cpu(), sb.cpu(), alpha=0.125)
    # the MFMA's internal 128-deep dot product is not an IEEE fp32 sum: measured 2.3e-5 of the output scale
    assert float((out - want).abs().max()) <= 1e-4 * float(want.abs().max())
    full = 0.125 * (A.double() @ B.double().T)
    rel = float((out - full).norm() / full.norm())
    print("MXFP8 GEMM vs fp64 product of the unquantised operands: rel-L2", rel)
    assert rel < 5e-2
    o16 = H.op_gemm_mx8(qa, sa, qb, sb, epilogue=0, alpha=0.125).float().cpu().double()
    assert float((o16 - want).abs().max()) <= 2 ** -8 * float(want.abs().max()) * 1.01


@pytest.mark.gpu
def test_hip_gemm_mx8_rejects_bad_shapes():
    from rald_amd import _handles as H
    q, s = H.op_quantize_mx8(torch.zeros(64, 96, device="cuda"))
    with pytest.raises(RuntimeError):
        H.op_gemm_mx8(q, s, q, s)                                     # K = 96: not a multiple of 128
    with pytest.raises(RuntimeError):
        H.op_quantize_mx8(torch.zeros(4, 48, device="cuda"))          # K % 32 != 0
    with pytest.raises(RuntimeError):
        H.op_quantize_mx8(torch.zeros(4, 64))                         # CPU tensor


@pytest.mark.gpu
def test_fp8_qkv_mode_vs_reference_goldens():
    """BASELINE config #5: the denoiser with MXFP8 q/k/v projections (qkv_dtype='fp8') against the SAME
    fp32 goldens as the bf16 mode (SURVEY.md §8d 'fp8: same three numbers').  Stated tolerances for this mode:
    one NFE D_x rel-L2 <= 7e-2 (measured 3.5e-2 / 3.0e-2 / 7e-5 at sigma 80 / 1 / 0.002), 18-step sampler
    <= 6e-2 (measured 2.9e-2) - bounds = 2x the measured values -, raw F_x within 1.3e-1 of the bf16 mode (measured
    6.5e-2 on these random weights).  e4m3 keeps 3 mantissa bits: ~3.8 % per projection (test_hip_gemm_mx8_*), 72 fp8 projections per NFE."""
    from conftest import load_golden, rel_l2
    from rald_amd import config, models_radar_generation as G, weights
    m = G.LatentArrayTransformer(in_channels=32, t_channels=256, n_heads=8, d_head=64, depth=24)
    m.load_state_dict(weights.make_state_dict(weights.dit_spec(depth=24, with_radar=False, prefix=""), 0), strict=True)
    m = m.cuda()
    x, t, cond = synth.latents([0, 1]).cuda(), torch.tensor([0.25, -1.0]), synth.cond_tokens(2).cuda()
    out16 = m(x, t, cond=cond)
    m.qkv_dtype = "fp8"
    out8 = m(x, t, cond=cond)
    print("fp8 vs bf16 mode, raw F_x rel_l2:", rel_l2(out8, out16))
    assert 1e-4 < rel_l2(out8, out16) < 1.3e-1                        # the mode really changes the arithmetic
    edm = G.EDMPrecond(n_latents=512, channels=32, depth=24, configs=config.shipped_generation_config())
    edm.load_state_dict(weights.make_state_dict(weights.dit_spec(depth=24), 0), strict=True)
    edm = edm.cuda()
    edm.qkv_dtype = "fp8"
    g3, g4 = load_golden("g3_precond.npz"), load_golden("g4_sample18.npz")
    cube = synth.radar_cube(2).cuda()
    for s in (80.0, 1.0, 0.002):
        err = rel_l2(edm(x * max(s, 1.0), torch.tensor(s), cube, "radar"), g3[f"d_sigma_{s}"])
        print(f"fp8 mode, sigma {s}: D_x rel_l2 {err}")
        assert err < 7e-2
    smp = edm.sample(cond=cube, batch_seeds=None, cond_type="radar")
    err = rel_l2(smp, g4["sample"])
    print("fp8 mode, 18-step sampler rel_l2", err)
    assert err < 6e-2
    with pytest.raises(ValueError):
        m.qkv_dtype = "int4"
        m(x, t, cond=cond)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["fp8_ff1", "fp8_ff"])
def test_fp8_ff_modes_vs_reference_goldens(mode):
    """qkv_dtype='fp8_ff1': MXFP8 q/k/v AND the GEGLU projection (its input, the norm3 output, also comes straight from the
    fused LayerNorm epilogue); 'fp8_ff': ff.net.2 as well (the GEGLU epilogue emits MXFP8, the fused residual+LayerNorm GEMM
    consumes it; B >= 32 - below that it behaves like 'fp8_ff1').  Same goldens; stated tolerances (2x the measured 4.6e-2 /
    3.8e-2 / 1e-4 per NFE and 3.7e-2 after the sampler): one NFE <= 9e-2, 18-step sampler <= 7.5e-2; a B = 32 NFE of 'fp8_ff'
    (the batch where its own kernels run) within 9.5e-2 of the bf16 mode (measured 4.7e-2)."""
    from conftest import load_golden, rel_l2
    from rald_amd import config, models_radar_generation as G, weights
    edm = G.EDMPrecond(n_latents=512, channels=32, depth=24, configs=config.shipped_generation_config())
    edm.load_state_dict(weights.make_state_dict(weights.dit_spec(depth=24), 0), strict=True)
    edm = edm.cuda()
    if mode == "fp8_ff":
        xb, cb = synth.latents(range(32)).cuda(), synth.radar_cube(2).cuda().repeat(16, 1, 1, 1, 1)
        ref = edm(xb, torch.tensor(1.0), cb, "radar")
        edm.qkv_dtype = mode
        err = rel_l2(edm(xb, torch.tensor(1.0), cb, "radar"), ref)
        print("fp8_ff vs bf16 mode at B=32, D_x rel_l2", err)
        assert 1e-4 < err < 9.5e-2
    edm.qkv_dtype = mode
    g3, g4 = load_golden("g3_precond.npz"), load_golden("g4_sample18.npz")
    cube, x = synth.radar_cube(2).cuda(), synth.latents([0, 1]).cuda()
    for s in (80.0, 1.0, 0.002):
        err = rel_l2(edm(x * max(s, 1.0), torch.tensor(s), cube, "radar"), g3[f"d_sigma_{s}"])
        print(f"{mode} mode, sigma {s}: D_x rel_l2 {err}")
        assert err < 9e-2
    err = rel_l2(edm.sample(cond=cube, batch_seeds=None, cond_type="radar"), g4["sample"])
    print(f"{mode} mode, 18-step sampler rel_l2", err)
    assert err < 7.5e-2
