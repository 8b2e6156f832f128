"""A/B of the two-workgroups-per-CU residual+LayerNorm GEMM (64-row tiles, 32-deep k-steps) against the 128-row kernel
(probe library: RALD_LIB_OVERRIDE=rald_amd/librald_hip_probe.so, switch RALD_LN_RING) - stand-alone launches and whole NFEs."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rald_amd import _handles as H, models_radar_generation as G, weights, synth
assert "probe" in os.environ.get("RALD_LIB_OVERRIDE", ""), "run with RALD_LIB_OVERRIDE=rald_amd/librald_hip_probe.so"

def timed(f, reps):
    for _ in range(3): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
M = B * 512
for K in (512, 2048):
    A = torch.randn(M, K, device="cuda").bfloat16()
    W = (torch.randn(512, K, device="cuda") / K ** 0.5).bfloat16()
    bias = torch.randn(512, device="cuda")
    g = torch.randn(B, 1024, device="cuda") * 0.1
    gs, bs = g[:, :512], g[:, 512:]
    outs = {}
    for rnd in range(3):
        line = f"K={K} B={B} round {rnd}: "
        for v in ("0", "1"):
            os.environ["RALD_LN_RING"] = v
            x = torch.zeros(M, 512, device="cuda")
            us = timed(lambda: H.op_gemm_resid_ln(A, W, bias, x, gs, bs, gstride=1024, rows_per_group=512, add_one=1.0), 20) * 1e3
            x = torch.ones(M, 512, device="cuda")
            h = H.op_gemm_resid_ln(A, W, bias, x, gs, bs, gstride=1024, rows_per_group=512, add_one=1.0)
            outs[v] = (x.clone(), h.float())
            line += f"ring={v} {us:6.1f} us ({2.0*M*512*K/us/1e6:5.0f} TF) | "
        dx = float((outs["0"][0] - outs["1"][0]).norm() / outs["0"][0].norm()); dh = float((outs["0"][1] - outs["1"][1]).norm() / outs["0"][1].norm())
        print(line + f"x diff {dx:.1e} h diff {dh:.1e}", flush=True)
m = G.LatentArrayTransformer(in_channels=32, t_channels=256, n_heads=8, d_head=64, depth=24)
m.load_state_dict(weights.make_state_dict(weights.dit_spec(depth=24, with_radar=False, prefix=""), 0)); m = m.cuda()
h = m._handle(512, 64); h.set_sigmas([1.0])
x = synth.latents(range(B)).cuda(); cache = h.encode_cond_tokens(synth.cond_tokens(B).cuda())
for rnd in range(3):
    line = f"NFE B={B} round {rnd}: "
    for v in ("0", "1"):
        os.environ["RALD_LN_RING"] = v
        ms = timed(lambda: h.denoise(x, cache, 0), 10)
        line += f"ring={v} {ms:7.3f} ms | "
    print(line, flush=True)
