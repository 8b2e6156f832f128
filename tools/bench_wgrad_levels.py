"""Conv3d weight gradient at every level of the radar encoder (B = 8), probe library: with / without the final atomics, forced line-range counts."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rald_amd._lib import lib, check
L = lib(); p = lambda t: C.c_void_p(t.data_ptr() if t is not None else 0)
def timed(B, D, H, W, Cin, Cout, reps=5, ws_form=False):
    x = torch.randn(B, D, H, W, Cin, device="cuda").bfloat16(); dy = torch.randn(B * D * H * W, Cout, device="cuda").bfloat16()
    dW = torch.zeros(Cout, Cin, 27, device="cuda"); db = torch.zeros(Cout, device="cuda")
    nb = L.rald_op_conv3d_wgrad_workspace_bytes(B, D, H, W, Cin, Cout, 1, 1)
    ws = torch.empty(max(nb, 16), device="cuda", dtype=torch.uint8)
    def run():
        if ws_form: check(L.rald_op_conv3d_wgrad_ws(p(dy), p(x), p(dW), p(db), B, D, H, W, Cin, Cout, 1, 1, p(ws), nb, None))
        else: check(L.rald_op_conv3d_wgrad(p(dy), p(x), p(dW), p(db), B, D, H, W, Cin, Cout, 1, 1, None))
    for _ in range(2): run()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): run()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3
levels = [(128, 64, 32, 64, 64), (64, 32, 16, 64, 64), (32, 16, 8, 64, 128), (32, 16, 8, 128, 128), (16, 8, 4, 128, 128), (8, 4, 2, 128, 256), (8, 4, 2, 256, 256)]
for (D, H, W, Cin, Cout) in levels:
    for sp, ab, wsf in ((0, 0, False), (0, 8, False), (0, 0, True), (16, 0, True), (32, 0, True), (64, 0, True)):
        os.environ["RALD_WGRAD_SPLITS"] = str(sp)
        os.environ["RALD_WGRAD_ABLATE"] = str(ab)
        us = timed(8, D, H, W, Cin, Cout, ws_form=wsf)
        print(f"{D}x{H}x{W} {Cin}->{Cout} splits={sp} ablate={ab:2d} workspace={int(wsf)}: {us:8.1f} us  ({2.0*8*D*H*W*Cin*Cout*27/us/1e6:6.0f} TFLOP/s)", flush=True)
# the transformer's Linear weight gradients (M = 4096 rows)
from rald_amd._handles import op_gemm_tn
for (N1, N2) in ((512, 2048), (4096, 512), (512, 512), (1536, 512)):
    A, Bm = torch.randn(4096, N1, device="cuda").bfloat16(), torch.randn(4096, N2, device="cuda").bfloat16()
    Cm, cs = torch.zeros(N1, N2, device="cuda"), torch.zeros(N1, device="cuda")
    for atomics in (True, False):
        for _ in range(2): op_gemm_tn(A, Bm, Cm, cs, atomics=atomics)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10): op_gemm_tn(A, Bm, Cm, cs, atomics=atomics)
        e.record(); torch.cuda.synchronize()
        us = s.elapsed_time(e) / 10 * 1e3
        print(f"gemm_tn M=4096 N1={N1} N2={N2} atomics={int(atomics)}: {us:7.1f} us ({2.0*4096*N1*N2/us/1e6:5.0f} TFLOP/s)", flush=True)
