import sys, math, torch
sys.path.insert(0, ".")
from rald_amd import _handles as H
def run(B, nq, nk, heads=8, reps=20):
    HD = heads*64; nkp=(nk+63)//64*64
    q=torch.randn(B,nq,HD,device="cuda").bfloat16(); k=torch.randn(B,nkp,HD,device="cuda").bfloat16()
    vt=torch.randn(B,HD,nkp,device="cuda").bfloat16()
    f=lambda: H.op_attention(q,k,vt,nk,heads,0.125)
    for _ in range(3): f()
    torch.cuda.synchronize(); s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): f()
    e.record(); torch.cuda.synchronize()
    us=s.elapsed_time(e)/reps*1e3
    return us, 4.0*B*heads*nq*nk*64/us/1e6
for B in (32, 64):
    for nk in (512, 64):
        us, tf = run(B, 512, nk)
        print(f"B={B} nq=512 nk={nk}: {us:7.1f} us  {tf:6.0f} TF", flush=True)
us, tf = run(8, 512, 10000); print(f"B=8 nq=512 nk=10000: {us:7.1f} us {tf:6.0f} TF")
