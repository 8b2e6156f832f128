"""Scratch: are the small-batch fused kernels bit-stable when other streams keep the chip busy?"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rald_amd._lib import check, lib
from rald_amd import _handles as H
c = lambda t: C.c_void_p(t.data_ptr())
g = torch.Generator("cpu").manual_seed(3)
D, NL, B = 512, 512, 2
qkv = (torch.randn(B * NL, 3 * D, generator=g) * 0.8).cuda().bfloat16()
Wo = (torch.randn(D, D, generator=g) / 22).cuda().bfloat16()
hin = torch.randn(B * NL, D, generator=g).cuda().bfloat16()
Kc = torch.randn(B * 64, 2 * D, generator=g).cuda().bfloat16()
Vt = torch.randn(B, 2 * D, 64, generator=g).cuda().bfloat16()
big_a = torch.randn(8192, 512, device="cuda").bfloat16(); big_b = torch.randn(4096, 512, device="cuda").bfloat16()
side = torch.cuda.Stream()
def self_proj():
    part = torch.empty(8, B * NL, D, device="cuda")
    check(lib().rald_op_attn_self_proj(c(qkv), 3 * D, c(Wo), c(part), NL, 8, B, C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    return part
def xattn():
    part = torch.empty(8, B * NL, D, device="cuda")
    check(lib().rald_op_xattn_q2_proj(c(hin), c(Wo), c(Kc), 2 * D, 64 * 2 * D, c(Vt), 64, 2 * D * 64, c(Wo), c(part), B * NL, NL, 8, 64, 0.18, C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    return part
for name, fn in (("self_proj", self_proj), ("xattn", xattn)):
    ref = fn(); torch.cuda.synchronize()
    bad = 0
    for it in range(30):
        with torch.cuda.stream(side):
            for _ in range(4): H.op_gemm_nt(big_a, big_b, epilogue=0)
            o2 = fn()
        o = fn()
        torch.cuda.synchronize()
        if not torch.equal(o, ref) or not torch.equal(o2, ref): bad += 1
    print(name, "mismatching rounds under concurrency:", bad, flush=True)
os.system(f"{sys.executable} {os.path.dirname(os.path.abspath(__file__))}/debug_concurrent.py")
