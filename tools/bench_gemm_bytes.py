"""Probe library: the FF1 / q|k|v GEMMs of the denoiser at B = 64 with parts of the stage DMA dropped (RALD_GEMM_ABLATE bits: 1 no DMA after the
prologue, 4096 no B pieces, 8192 no A pieces, 2 no epilogue) - is a k-step paced by the bytes of its stage or by a stage's latency?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rald_amd import _handles as H
def run(M, N, K, epi, ab, reps=30):
    os.environ["RALD_GEMM_ABLATE"] = str(ab)
    A = torch.randn(M, K, device="cuda").bfloat16(); W = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16(); bias = torch.randn(N, device="cuda")
    f = lambda: H.op_gemm_nt(A, W, bias=bias, epilogue=epi)
    for _ in range(5): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3
for name, M, N, K, epi in (("ff1 geglu", 32768, 4096, 512, 3), ("ff1-shaped K=2048", 32768, 4096, 2048, 3), ("qkv", 32768, 1536, 512, 0)):
    for rnd in range(2):
        line = f"{name} M={M} N={N} K={K}:"
        for ab in (0, 4096, 8192, 4096 + 8192, 1, 2, 2 + 4096 + 8192):
            line += f"  ab{ab}: {run(M, N, K, epi, ab):7.1f}us |"
        print(line, flush=True)
