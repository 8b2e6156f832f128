"""Scratch microbench: MXFP8 GEMM vs bf16 GEMM on the denoiser's QKV / proj shapes."""
import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rald_amd import _handles as H

def timeit(f, reps=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3

Bs = [int(b) for b in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["32", "64"])]
for B in Bs:
    M = B * 512
    for name, N, K, epi in [("qk", 1024, 512, 0), ("qkv", 1536, 512, 0), ("q", 512, 512, 0), ("proj resid", 512, 512, 2), ("ff1-like", 4096, 512, 0)]:
        A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") / K ** 0.5
        A16, W16 = A.bfloat16(), W.bfloat16()
        qa, sa = H.op_quantize_mx8(A16); qb, sb = H.op_quantize_mx8(W)
        x = torch.zeros(M, N, device="cuda") if epi == 2 else None
        t16 = timeit(lambda: H.op_gemm_nt(A16, W16, epilogue=epi, C_inout=x))
        t8 = timeit(lambda: H.op_gemm_mx8(qa, sa, qb, sb, epilogue=epi, C_inout=x))
        tq = timeit(lambda: H.op_quantize_mx8(A16))
        fl = 2.0 * M * N * K
        print(f"B={B:3d} {name:11s} M={M} N={N} K={K}: bf16 {t16:7.1f}us {fl/t16/1e6:6.0f}TF | mx8 {t8:7.1f}us {fl/t8/1e6:6.0f}TF | quantise A {tq:6.1f}us", flush=True)
