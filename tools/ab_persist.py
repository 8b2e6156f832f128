"""A/B of the persistent GEGLU GEMM (probe library: RALD_LIB_OVERRIDE=rald_amd/librald_hip_probe.so) - stand-alone FF1 launches
and whole NFEs at B = 64, interleaved rounds in one process."""
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

variants = [("plain", "0", "0"), ("persist", "1", "0"), ("persist+stagger", "1", "1")]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
M = B * 512
A = torch.randn(M, 512, device="cuda").bfloat16(); W = (torch.randn(4096, 512, device="cuda") / 22).bfloat16(); bias = torch.randn(4096, device="cuda")
ref = None
for rnd in range(3):
    line = f"FF1 alone B={B} round {rnd}: "
    for name, p, s in variants:
        os.environ["RALD_GEMM_PERSIST"], os.environ["RALD_GEMM_STAGGER"] = p, s
        ms = timed(lambda: H.op_gemm_nt(A, W, bias=bias, epilogue=3), 30)
        out = H.op_gemm_nt(A, W, bias=bias, epilogue=3)
        if ref is None: ref = out
        line += f"{name} {ms*1e3:6.1f} us ({2.0*M*4096*512/ms/1e9:5.0f} TF, equal {bool(torch.equal(out, ref))}) | "
    print(line, flush=True)
m = G.LatentArrayTransformer(in_channels=32, t_channels=256, n_heads=8, d_head=64, depth=24)
m.load_state_dict(weights.make_state_dict(weights.dit_spec(depth=24, with_radar=False, prefix=""), 0)); m = m.cuda()
h = m._handle(512, 64); h.set_sigmas([1.0])
x = synth.latents(range(B)).cuda(); cache = h.encode_cond_tokens(synth.cond_tokens(B).cuda())
for rnd in range(3):
    line = f"NFE B={B} round {rnd}: "
    for name, p, s in variants:
        os.environ["RALD_GEMM_PERSIST"], os.environ["RALD_GEMM_STAGGER"] = p, s
        ms = timed(lambda: h.denoise(x, cache, 0), 10)
        line += f"{name} {ms:7.3f} ms | "
    print(line, flush=True)
