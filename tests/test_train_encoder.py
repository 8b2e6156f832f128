"""Backward building blocks of the radar-spectrum encoder (models_radar_encoder.py:29-241)
against torch autograd on the CPU: Conv3d forward / data gradient / weight gradient (stride 1 and the Downsample
form F.pad(0,1) + k3 s2), GroupNorm(32, eps 1e-6)(+swish) forward / backward.  bf16 MFMA operands, fp32
accumulation: rel-L2 <= 1e-2 against fp32 math on the same (bf16-rounded where the kernel rounds) inputs."""
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_l2
from rald_amd import synth

pytestmark = pytest.mark.gpu


def _cl(t):      # NCDHW -> channels-last NDHWC contiguous
    return t.permute(0, 2, 3, 4, 1).contiguous()


def _cf(t):      # NDHWC -> NCDHW
    return t.permute(0, 4, 1, 2, 3).contiguous()


def test_conv3d_forward_dgrad_wgrad_vs_autograd():
    from rald_amd import train_encoder as TE
    for (B, D, H, W, Cin, Cout) in ((2, 8, 6, 4, 64, 128), (1, 4, 4, 8, 128, 16)):
        x = synth.normal([B, Cin, D, H, W], 600).bfloat16().float().requires_grad_()
        Wt = (synth.normal([Cout, Cin, 3, 3, 3], 601) / (Cin * 27) ** 0.5).requires_grad_()
        b = synth.normal([Cout], 602) * 0.1
        y = F.conv3d(x, Wt.detach().bfloat16().float() + (Wt - Wt.detach()), b, padding=1)     # forward on the bf16-rounded weights, gradient to Wt
        dy = synth.normal([B, Cout, D, H, W], 603)
        y.backward(dy)
        x16 = _cl(x.detach()).bfloat16().cuda()
        out = TE.conv3d(x16, TE.pack_conv(Wt.detach().cuda()), b.cuda())
        assert rel_l2(_cf(out.cpu()), y.detach()) < 3e-3
        dx = TE.conv_dgrad(_cl(dy).cuda(), Wt.detach().cuda())
        print("conv dgrad rel_l2", rel_l2(_cf(dx.cpu()), x.grad))
        assert rel_l2(_cf(dx.cpu()), x.grad) < 1e-2
        # bf16 dy in (Cout a multiple of 64: read as it lies) and the bf16 result the GroupNorm backward reads: the fp32 result rounded
        if Cout % 64 == 0:
            dx16 = TE.conv_dgrad(_cl(dy).bfloat16().cuda(), Wt.detach().cuda(), out_bf16=True)
            assert dx16.dtype == torch.bfloat16 and torch.equal(dx16, dx.bfloat16())
        dW, db = torch.zeros_like(Wt, device="cuda"), torch.zeros(Cout, device="cuda")
        TE.conv_wgrad(_cl(dy).cuda(), x16, dW, db)
        print("conv wgrad rel_l2", rel_l2(dW.cpu(), Wt.grad), rel_l2(db.cpu(), dy.sum((0, 2, 3, 4))))
        # (the bias gradient is the column sum of the bf16 copy of dy that feeds the MFMAs: ~1.5e-3, unbiased; it was an fp32 pass of its own)
        assert rel_l2(dW.cpu(), Wt.grad) < 1e-2 and rel_l2(db.cpu(), dy.sum((0, 2, 3, 4))) < 5e-3


@pytest.mark.parametrize("B,D,H,W", [(2, 32, 64, 32), (4, 64, 64, 32), (4, 32, 32, 16)])
def test_plane_staged_convolution_equals_the_line_staged_one_on_a_crop(B, D, H, W):
    """The 64-channel convolutions of large volumes run on the plane-staged kernels (csrc/radar.hip: conv3d_plane_kernel, and its persistent
    form conv3d_pplane_kernel from 256 columns x segments up - the second shape); small volumes stay on conv3d_line_kernel.  A convolution is
    local: the result on a d-slab of the big volume must equal the result on that slab cropped out with a one-plane halo - computed by the
    other kernel.  fp32 accumulation of the same 1 728 products per output in both: 1e-6; with the residual and as bf16 output too."""
    from rald_amd import train_encoder as TE
    Cc = 64
    x16 = synth.normal([B, D, H, W, Cc], 640).bfloat16().cuda()
    Wt = (synth.normal([Cc, Cc, 3, 3, 3], 641) / (Cc * 27) ** 0.5).cuda()
    bias = (synth.normal([Cc], 642) * 0.1).cuda()
    resid = synth.normal([B, D, H, W, Cc], 643).cuda()
    wp = TE.pack_conv(Wt)
    big = TE.conv3d(x16, wp, bias)
    big_r = TE.conv3d(x16, wp, bias, resid=resid)
    big_16 = TE.conv3d(x16, wp, bias, out_bf16=True)
    assert torch.equal(big_16, big.bfloat16())
    for b, d0 in ((0, 0), (B - 1, D - 8), (B // 2, D // 2 - 4)):
        lo, hi = max(d0 - 1, 0), min(d0 + 9, D)                      # the slab [d0, d0 + 8) with its halo planes where they exist
        crop = TE.conv3d(x16[b:b + 1, lo:hi].contiguous(), wp, bias)
        crop_r = TE.conv3d(x16[b:b + 1, lo:hi].contiguous(), wp, bias, resid=resid[b:b + 1, lo:hi].contiguous())
        sl = slice(d0 - lo, d0 - lo + 8)
        assert rel_l2(big[b, d0:d0 + 8].cpu(), crop[0, sl].cpu()) < 1e-6, (b, d0)
        assert rel_l2(big_r[b, d0:d0 + 8].cpu(), crop_r[0, sl].cpu()) < 1e-6, (b, d0)


def test_weight_gradients_through_the_workspace_are_bit_reproducible_and_match_the_atomic_form():
    """conv3d / Linear weight gradients with the row ranges meeting in a workspace (summed in order by a second launch) instead of
    fp32 atomics: two runs give identical bits, the result accumulates INTO the destination, and it equals the atomic form to fp32
    summation-order noise (1e-6).  Shapes: line-staged kernel (64 -> 128 at width 8, several 64 x 64 blocks and ranges), generic
    kernel in its convolution form (width 2) and as the plain row-contracting GEMM."""
    import ctypes as C
    from rald_amd import train_encoder as TE
    from rald_amd._handles import op_gemm_tn
    from rald_amd._lib import check, lib
    p = lambda t: C.c_void_p(t.data_ptr() if t is not None else 0)
    for (B, D, H, W, Cin, Cout) in ((4, 16, 8, 8, 64, 128), (2, 4, 4, 2, 128, 256)):
        x16 = synth.normal([B, D, H, W, Cin], 620).bfloat16().cuda()
        dy = synth.normal([B, D, H, W, Cout], 621).bfloat16().cuda()
        outs = []
        for _ in range(2):
            dW, db = torch.full((Cout, Cin, 3, 3, 3), 0.5, device="cuda"), torch.full((Cout,), 0.25, device="cuda")
            TE.conv_wgrad(dy, x16, dW, db)
            outs.append((dW, db))
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
        aW, ab = torch.full((Cout, Cin, 3, 3, 3), 0.5, device="cuda"), torch.full((Cout,), 0.25, device="cuda")
        check(lib().rald_op_conv3d_wgrad(p(dy), p(x16), p(aW), p(ab), B, D, H, W, Cin, Cout, 1, 1, None))
        torch.cuda.synchronize()
        assert rel_l2(outs[0][0].cpu(), aW.cpu()) < 1e-6 and rel_l2(outs[0][1].cpu(), ab.cpu()) < 1e-6
    A, Bm = synth.normal([1024, 256], 622).bfloat16().cuda(), synth.normal([1024, 384], 623).bfloat16().cuda()
    res = []
    for atomics in (False, False, True):
        Cm, cs = torch.ones(256, 384, device="cuda"), torch.ones(256, device="cuda")
        op_gemm_tn(A, Bm, Cm, cs, atomics=atomics)
        res.append((Cm, cs))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    assert rel_l2(res[0][0].cpu(), res[2][0].cpu()) < 1e-6 and rel_l2(res[0][1].cpu(), res[2][1].cpu()) < 1e-6
    want = A.float().T @ Bm.float() + 1
    assert rel_l2(res[0][0].cpu(), want.cpu()) < 1e-5


@pytest.mark.parametrize("B,D,H,W,Cc", [(2, 8, 4, 4, 64), (2, 8, 8, 16, 64)])      # output width 2: generic weight-gradient kernel; 8: the line-staged stride-2 form
def test_downsample_forward_dgrad_wgrad_vs_autograd(B, D, H, W, Cc):
    from rald_amd import train_encoder as TE
    x = synth.normal([B, Cc, D, H, W], 610).bfloat16().float().requires_grad_()
    Wt = (synth.normal([Cc, Cc, 3, 3, 3], 611) / (Cc * 27) ** 0.5).requires_grad_()
    b = synth.normal([Cc], 612) * 0.1
    y = F.conv3d(F.pad(x, (0, 1, 0, 1, 0, 1)), Wt.detach().bfloat16().float() + (Wt - Wt.detach()), b, stride=2)      # Downsample.forward :37-41
    dy = synth.normal(list(y.shape), 613)
    y.backward(dy)
    x16 = _cl(x.detach()).bfloat16().cuda()
    out = TE.conv3d(x16, TE.pack_conv(Wt.detach().cuda()), b.cuda(), stride=2, pad=0)
    assert out.shape == (B, D // 2, H // 2, W // 2, Cc)
    assert rel_l2(_cf(out.cpu()), y.detach()) < 3e-3
    dx = TE.down_dgrad(_cl(dy).cuda(), Wt.detach().cuda())
    print("downsample dgrad rel_l2", rel_l2(_cf(dx.cpu()), x.grad))
    assert rel_l2(_cf(dx.cpu()), x.grad) < 1e-2
    dW, db = torch.zeros_like(Wt, device="cuda"), torch.zeros(Cc, device="cuda")
    TE.conv_wgrad(_cl(dy).cuda(), x16, dW, db, stride=2, pad=0)
    print("downsample wgrad rel_l2", rel_l2(dW.cpu(), Wt.grad), rel_l2(db.cpu(), dy.sum((0, 2, 3, 4))))
    assert rel_l2(dW.cpu(), Wt.grad) < 1e-2 and rel_l2(db.cpu(), dy.sum((0, 2, 3, 4))) < 5e-3


def test_groupnorm_swish_forward_backward_vs_autograd():
    from rald_amd import train_encoder as TE
    for Cc, swish in ((64, True), (128, False), (256, True)):
        B, D, H, W = 2, 8, 4, 2
        x = (synth.normal([B, Cc, D, H, W], 620) * 1.7 + 0.4).requires_grad_()
        g, b = (1 + 0.1 * synth.normal([Cc], 621)).requires_grad_(), (0.1 * synth.normal([Cc], 622)).requires_grad_()
        y = F.group_norm(x, 32, g, b, eps=1e-6)
        a = y * torch.sigmoid(y) if swish else y
        da = synth.normal(list(a.shape), 623)
        a.backward(da)
        xc = _cl(x.detach()).cuda()
        y16, stats = TE.groupnorm(xc, g.detach().cuda(), b.detach().cuda(), swish)
        assert rel_l2(_cf(y16.float().cpu()), a.detach()) < 4e-3
        dx = torch.ones_like(xc)
        dg, db = torch.zeros(Cc, device="cuda"), torch.zeros(Cc, device="cuda")
        TE.groupnorm_bwd(xc, stats, g.detach().cuda(), b.detach().cuda(), _cl(da).cuda(), dx, dg, db, swish, accumulate=True)
        e = (rel_l2(_cf(dx.cpu()) - 1, x.grad), rel_l2(dg.cpu(), g.grad), rel_l2(db.cpu(), b.grad))
        print(f"groupnorm C={Cc} swish={swish}: dx {e[0]:.2e} dgamma {e[1]:.2e} dbeta {e[2]:.2e}")
        assert max(e) < 1e-4
        # the forms the encoder's backward uses: activations re-created from the saved statistics, dx also (or only) as bf16
        assert torch.equal(TE.groupnorm_apply(xc, stats, g.detach().cuda(), b.detach().cuda(), swish), y16)
        dx2, dx16 = torch.ones_like(xc), torch.zeros(xc.shape, device="cuda", dtype=torch.bfloat16)
        dg2, db2 = torch.zeros(Cc, device="cuda"), torch.zeros(Cc, device="cuda")
        TE.groupnorm_bwd(xc, stats, g.detach().cuda(), b.detach().cuda(), _cl(da).cuda(), dx2, dg2, db2, swish, accumulate=True, dx_bf16=dx16)
        # (no atomics anywhere in the backward: a second launch reproduces dx, dgamma and dbeta bit for bit)
        assert torch.equal(dx2, dx) and torch.equal(dg2, dg) and torch.equal(db2, db) and torch.equal(dx16, dx2.bfloat16())
        only16 = torch.zeros_like(dx16)
        TE.groupnorm_bwd(xc, stats, g.detach().cuda(), b.detach().cuda(), _cl(da).cuda(), None, dg2, db2, swish, accumulate=False, dx_bf16=only16)
        assert rel_l2(only16.float().cpu(), (dx - 1).cpu()) < 4e-3
        # da handed over as bf16 (a data-gradient convolution's bf16 result): the fp32 math on the rounded values
        da16 = _cl(da).bfloat16().cuda()
        ref = torch.zeros_like(xc)
        TE.groupnorm_bwd(xc, stats, g.detach().cuda(), b.detach().cuda(), da16.float(), ref, dg2, db2, swish, accumulate=False)
        got, got16 = torch.zeros_like(xc), torch.zeros_like(dx16)
        TE.groupnorm_bwd(xc, stats, g.detach().cuda(), b.detach().cuda(), da16, got, dg2, db2, swish, accumulate=False, dx_bf16=got16)
        assert rel_l2(got.cpu(), ref.cpu()) < 1e-6 and torch.equal(got16, got.bfloat16())


def test_encoder_and_tokeniser_forward_backward_vs_oracle_autograd():
    """EDMPrecond.process_radar_cond (:363-407: Encoder + radar_token_project + r/a/e embeddings) on a full-size cube
    (128 x 64 x 32 is the smallest the architecture's 64-token attention admits), B = 1: tokens and EVERY parameter
    gradient against autograd of the CPU oracle (itself pinned to the reference's encoder by G7 / G3)."""
    from oracle import rald_oracle as O
    from rald_amd import train_encoder as TE, weights
    sd = {k: v for k, v in weights.make_state_dict(weights.dit_spec(depth=1), 0).items() if k.startswith("radar_")}
    cube = synth.radar_cube(1)
    dtok = synth.normal([1, 64, 512], 630)
    leaf = {k: v.clone().requires_grad_() for k, v in sd.items()}
    tok_ref = O.process_radar_cond(leaf, cube)
    (tok_ref * dtok).sum().backward()
    params = {k: torch.nn.Parameter(v.clone().cuda()) for k, v in sd.items()}
    enc = TE.EncoderTrainer(params)
    tok = enc.forward(cube.cuda())
    e = rel_l2(tok.cpu(), tok_ref.detach())
    print("tokens rel_l2", e)
    assert e < 1.5e-2
    enc.backward(dtok.cuda())
    torch.cuda.synchronize()
    worst = ("", 0.0)
    for k, p in params.items():
        assert p.grad is not None, k
        if k.endswith(".k.bias"):
            # a bias on the keys shifts every score of a row equally: softmax is invariant, the true gradient is exactly 0
            # (autograd returns round-off); check that ours is small against the query-bias gradient of the same block
            qn = float(leaf[k.replace(".k.bias", ".q.bias")].grad.norm())
            assert float(leaf[k].grad.norm()) < 1e-4 * qn and float(p.grad.norm()) < 3e-2 * qn, k
            continue
        e = rel_l2(p.grad.cpu(), leaf[k].grad)
        if e > worst[1]:
            worst = (k, e)
        assert e < 5e-2, (k, e)
    print("worst encoder parameter-gradient rel_l2:", worst)


def test_full_edm_training_step_vs_oracle_autograd_and_loss_decreases():
    """The reference's training iteration with the radar encoder trained jointly (engine_generation.py:74-110 at the shipped
    unfreeze_radar_enc = true): cube -> encoder -> tokens -> denoiser (depth 2 here) -> EDMLoss.  Loss and every one of the
    model's parameter gradients against autograd of the CPU oracle; then three optimizer steps lower the loss."""
    from oracle import rald_oracle as O
    from rald_amd import config, models_radar_generation as G, train_dit as TD, weights
    from rald_amd.train_utils import FlatAdamW
    depth, Bn = 2, 1
    sd = weights.make_state_dict(weights.dit_spec(depth=depth), 0)
    cube, y = synth.radar_cube(Bn), synth.normal([Bn, 512, 32], 640)
    rnd, noise = synth.normal([Bn, 1, 1], 641), synth.normal([Bn, 512, 32], 642)
    leaf = {k: v.clone().requires_grad_() for k, v in sd.items()}
    loss_ref = O.edm_loss(leaf, y, O.process_radar_cond(leaf, cube), rnd, noise, depth)
    loss_ref.backward()
    m = G.EDMPrecond(n_latents=512, channels=32, depth=depth, configs=config.shipped_generation_config())
    m.load_state_dict(sd, strict=True)
    m = m.cuda()
    opt = FlatAdamW(list(m.parameters()), lr=1e-4, ema=True)
    tr = TD.EdmTrainer(m, opt)
    opt.zero_grad()
    loss = tr.forward_backward(y.cuda(), cube.cuda(), rnd.flatten(), noise.cuda())
    torch.cuda.synchronize()
    print("EDM loss (encoder trained jointly) hip / oracle:", float(loss), float(loss_ref.detach()))
    assert abs(float(loss) - float(loss_ref.detach())) < 5e-3 * float(loss_ref.detach())
    worst = ("", 0.0)
    for k, p in m.named_parameters():
        ref = leaf[k].grad
        if k.endswith(".k.bias"):
            continue                                              # exactly-zero gradient: see the encoder test
        e = rel_l2(p.grad.cpu(), ref)
        if e > worst[1]:
            worst = (k, e)
        assert e < 6e-2, (k, e)
    print("worst parameter-gradient rel_l2 over the whole model:", worst)
    losses = [float(loss)]
    for _ in range(3):
        l, _ = tr.step(y.cuda(), cube.cuda(), rnd.flatten(), noise.cuda())
        losses.append(float(l))
    print("losses:", [round(v, 4) for v in losses])
    assert losses[-1] < losses[0]
