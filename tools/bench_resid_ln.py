"""Scratch: fused residual-GEMM + LayerNorm (gemm_resid_ln) vs EPI_RESID GEMM + separate LayerNorm."""
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

for B in [int(b) for b in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["16", "32", "64"])]:
    M = B * 512
    for K in (512, 2048):
        A = torch.randn(M, K, device="cuda").bfloat16()
        W = (torch.randn(512, K, device="cuda") / K ** 0.5).bfloat16()
        bias = torch.randn(512, device="cuda")
        x = torch.zeros(M, 512, device="cuda")
        g = torch.randn(B, 1024, device="cuda") * 0.1
        gs, bs = g[:, :512], g[:, 512:]
        fused = timeit(lambda: H.op_gemm_resid_ln(A, W, bias, x, gs, bs, gstride=1024, rows_per_group=512, add_one=1.0))
        t_g = timeit(lambda: H.op_gemm_nt(A, W, bias=bias, epilogue=2, C_inout=x))
        t_l = timeit(lambda: H.op_layernorm(x, gs, bs, gstride=1024, rows_per_group=512, add_one=1.0))
        print(f"B={B:3d} K={K:4d}: fused {fused:7.1f}us | gemm {t_g:7.1f} + ln {t_l:6.1f} = {t_g + t_l:7.1f}us", flush=True)
