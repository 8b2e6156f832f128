"""The denoiser's large GEMM shapes at B = 64 through variant libraries (tools/build_variant.sh), one subprocess per library and round.
usage: python tools/ab_libs_gemm.py <rounds> <lib> [<lib> ...]"""
import os, subprocess, sys
child = r'''
import os, sys, torch
sys.path.insert(0, os.environ["RALD_ROOT"])
from rald_amd import _handles as H
def run(M, N, K, epi, reps=40):
    A = torch.randn(M, K, device="cuda").bfloat16(); W = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16(); bias = torch.randn(N, device="cuda")
    f = lambda: H.op_gemm_nt(A, W, bias=bias, epilogue=epi)
    for _ in range(5): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3
print("  ".join(f"{n} {run(M, N, K, epi):6.1f}us" for n, M, N, K, epi in (("ff1", 32768, 4096, 512, 3), ("ff1-K2048", 32768, 4096, 2048, 3), ("qkv", 32768, 1536, 512, 0))))
'''
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rounds, libs = int(sys.argv[1]), sys.argv[2:]
for r in range(rounds):
    for lib in libs:
        env = dict(os.environ, RALD_ROOT=root, RALD_LIB_OVERRIDE=os.path.join(root, lib))
        out = subprocess.run([sys.executable, "-c", child], env=env, capture_output=True, text=True, timeout=300)
        print(f"{os.path.basename(lib):28s} {out.stdout.strip() or out.stderr.strip()[-300:]}", flush=True)
