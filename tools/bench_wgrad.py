"""Conv3d weight gradient (line-staged kernel) at the encoder's full-resolution shape, with the probe library's ablation bits."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rald_amd._lib import lib, check
L = lib(); p = lambda t: C.c_void_p(t.data_ptr() if t is not None else 0)
B, D, H, W, Cin, Cout = 8, 128, 64, 32, 64, 64
x = torch.randn(B, D, H, W, Cin, device="cuda").bfloat16(); dy = torch.randn(B * D * H * W, Cout, device="cuda").bfloat16()
dW = torch.zeros(Cout, Cin, 27, device="cuda"); db = torch.zeros(Cout, device="cuda")
def timed(reps=5):
    for _ in range(2): check(L.rald_op_conv3d_wgrad(p(dy), p(x), p(dW), p(db), B, D, H, W, Cin, Cout, 1, 1, None))
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): check(L.rald_op_conv3d_wgrad(p(dy), p(x), p(dW), p(db), B, D, H, W, Cin, Cout, 1, 1, None))
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3
for sp, ab in ((0, 0), (56, 0), (112, 0), (24, 0), (56, 8), (112, 8)):
    os.environ["RALD_WGRAD_SPLITS"] = str(sp)
    os.environ["RALD_WGRAD_ABLATE"] = str(ab)
    us = timed()
    print(f"splits={sp} ablate={ab:2d}: {us:8.1f} us  ({2.0*B*D*H*W*Cin*Cout*27/us/1e6:6.0f} TFLOP/s nominal)", flush=True)
