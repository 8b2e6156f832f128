"""Row-contracting GEMM (Linear weight gradients at 4 096 rows), workspace form: how many row ranges (probe library, RALD_TN_TARGET workgroups)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rald_amd._handles import op_gemm_tn
for (N1, N2) in ((512, 2048), (4096, 512), (512, 512), (1536, 512)):
    A, Bm = torch.randn(4096, N1, device="cuda").bfloat16(), torch.randn(4096, N2, device="cuda").bfloat16()
    Cm, cs = torch.zeros(N1, N2, device="cuda"), torch.zeros(N1, device="cuda")
    for atomics, target in ((True, 1024), (False, 1024), (False, 512), (False, 256), (True, 512)):
        os.environ["RALD_TN_TARGET"] = str(target)
        for _ in range(2): op_gemm_tn(A, Bm, Cm, cs, atomics=atomics)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10): op_gemm_tn(A, Bm, Cm, cs, atomics=atomics)
        e.record(); torch.cuda.synchronize()
        us = s.elapsed_time(e) / 10 * 1e3
        print(f"gemm_tn M=4096 N1={N1} N2={N2} atomics={int(atomics)} target={target}: {us:7.1f} us ({2.0*4096*N1*N2/us/1e6:5.0f} TFLOP/s)", flush=True)
