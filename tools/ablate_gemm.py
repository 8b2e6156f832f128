import os, sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rald_amd import _handles as H
def run(M,N,K,epi,impl,abl,reps=20):
    os.environ["RALD_GEMM_IMPL"]=str(impl); os.environ["RALD_GEMM_ABLATE"]=str(abl)
    A=torch.randn(M,K,device="cuda").bfloat16(); W=(torch.randn(N,K,device="cuda")/K**0.5).bfloat16(); b=torch.randn(N,device="cuda")
    x=torch.zeros(M,N,device="cuda") if epi==2 else None
    f=lambda: H.op_gemm_nt(A,W,bias=b,epilogue=epi,C_inout=x)
    for _ in range(3): f()
    torch.cuda.synchronize(); s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)/reps*1e3
if __name__ != "__main__": raise SystemExit
M=16384
if len(sys.argv) > 1 and sys.argv[1] == "prio":
    for name,N,K,epi in [("ff1",4096,512,3),("qk",1024,512,0),("ff2",512,2048,2)]:
        for impl in (1, 3, 5):
            print(name, "impl%d" % impl, " ".join(f"abl{a}:{run(M,N,K,epi,impl,a):7.1f}us" for a in (0, 32, 0, 32)), flush=True)
    raise SystemExit
if len(sys.argv) > 1 and sys.argv[1] == "nt":
    for M in (16384, 32768):
        for name,N,K,epi in [("ff1",4096,512,3),("qkv",1536,512,0),("q",512,512,0)]:
            print(M, name, " ".join(f"abl{a}:{run(M,N,K,epi,5,a):7.1f}us" for a in (0, 64, 0, 64)), flush=True)
    raise SystemExit
if len(sys.argv) > 1 and sys.argv[1] == "epi":
    for impl in (1, 5):
        print("ff1 impl%d" % impl, " ".join(f"abl{a}:{run(M,4096,512,3,impl,a):7.1f}us" for a in (0, 8, 16, 24, 2)), flush=True)
    raise SystemExit
if len(sys.argv) > 1:
    for name,N,K,epi in [("ff1",4096,512,3),("qk",1024,512,0),("ff2",512,2048,2)]:
        print(name, "impl1 stagger sleeps:", " ".join(f"{n}:{run(M,N,K,epi,1,4+16*n):6.1f}us" for n in (0,1,2,3,4,6)), flush=True)
    raise SystemExit
for name,N,K,epi in [("ff1",4096,512,3),("ff1-bf16epi",4096,512,0),("ff2",512,2048,2),("qk",1024,512,0)]:
    for impl in (1,5):
        print(name, f"impl{impl}", " ".join(f"abl{a}:{run(M,N,K,epi,impl,a):7.1f}us" for a in (0,1,2,3)), flush=True)
