"""Scratch microbench: the denoiser's GEMM shapes through rald_op_gemm_nt, interleaved A/B of
kernel variants (env RALD_GEMM_IMPL is read per launch by the library)."""
import os, sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rald_amd import _handles as H

def run(M, N, K, epi, impl, reps=20):
    os.environ["RALD_GEMM_IMPL"] = str(impl)
    A = torch.randn(M, K, device="cuda").bfloat16()
    W = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16()
    bias = torch.randn(N, device="cuda")
    x = torch.zeros(M, N, device="cuda") if epi == 2 else None
    f = lambda: H.op_gemm_nt(A, W, bias=bias, epilogue=epi, C_inout=x)
    for _ in range(3): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): f()
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / reps
    return ms, 2.0 * M * N * K / ms / 1e9

impls = [int(i) for i in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["0", "1"])]
Bs = [int(b) for b in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["32"])]
shapes = [("ff1 geglu", 4096, 512, 3), ("ff2 resid", 512, 2048, 2), ("qk bf16", 1024, 512, 0), ("proj resid", 512, 512, 2)]
for B in Bs:
    M = B * 512
    for name, N, K, epi in shapes:
        line = f"B={B:3d} {name:11s} M={M} N={N} K={K}: "
        for rnd in range(2):
            for impl in impls:
                ms, tf = run(M, N, K, epi, impl)
                line += f" impl{impl}: {ms*1e3:7.1f}us {tf:6.0f}TF |"
        print(line, flush=True)
