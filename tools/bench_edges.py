"""First / last layer of the denoiser at B = 64 (proj_in, final_norm_proj): stand-alone timings."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rald_amd._lib import lib, check
L = lib(); p = lambda t: C.c_void_p(t.data_ptr())
def timed(f, reps=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3
for B in (16, 64):
    M = B * 512
    xin = torch.randn(M, 32, device="cuda"); W = torch.randn(512, 32, device="cuda"); coef = torch.rand(B, 4, device="cuda") + 0.5
    x = torch.randn(M, 512, device="cuda"); g = torch.ones(512, device="cuda"); b = torch.zeros(512, device="cuda"); Wo = torch.randn(32, 512, device="cuda")
    out = torch.empty(M, 32, device="cuda")
    t1 = timed(lambda: check(L.rald_op_proj_in(p(xin), p(W), p(x), M, 32, 512, p(coef), 4, 512, None)))
    t2 = timed(lambda: check(L.rald_op_final_norm_proj(p(x), p(g), p(b), p(Wo), p(xin), p(out), M, 512, 32, p(coef), 4, 512, None)))
    print(f"B={B}: proj_in {t1:.1f} us ({M*512*4/t1/1e6:.2f} TB/s written), final_norm_proj {t2:.1f} us ({M*512*4/t2/1e6:.2f} TB/s read)")
