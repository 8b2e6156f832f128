"""Streaming query decoder: time per launch for the waves-per-workgroup variants, queries/s, and the error against G5."""
import ctypes as C
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from rald_amd import models_ae as A, synth, weights
from rald_amd._lib import lib, check

m = A.kl_d512_m512_l32_mix(N=10000)
m.load_state_dict(weights.make_state_dict(weights.spec_of_state_dict(m.state_dict()), 0), strict=True)
m = m.cuda()
h = m._handle()
g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "g5_ae.npz"))
z = torch.from_numpy(g["z"]).cuda()
ctx = h.decode_latents(z)
q = synth.queries(2, 4096).cuda()
out = h.decode_queries(ctx, q).cpu()
ref = torch.from_numpy(g["logits"]).squeeze(-1)
print("G5 decode logits rel_l2", float((out - ref).norm() / ref.norm()), "max abs", float((out - ref).abs().max()))

def run(ctx1, qq, nw):
    B, Q, _ = qq.shape
    o = torch.empty(B, Q, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    f = lambda: check(lib().rald_op_ae_decode_queries_nw(h._h, C.c_void_p(ctx1.data_ptr()), C.c_void_p(qq.data_ptr()), B, Q, C.c_void_p(o.data_ptr()), nw, C.c_void_p(st)))
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 10, o

ctx1 = h.decode_latents(z[:1].contiguous())
for Q in (10000, 500000, 1200000):
    qq = synth.queries(1, Q, seed=3).cuda()
    base = None
    for nw in (8, 12, 16):
        ms, o = run(ctx1, qq, nw)
        if base is None: base = o
        print(f"Q={Q} nw={nw}: {ms*1e3:.1f} us  ({Q/ms/1e3:.1f} M queries/s)  same-as-nw8 {bool(torch.equal(o, base))}")
ctx8 = h.decode_latents(synth.normal([8, 512, 32], 3).cuda())
q8 = synth.queries(8, 10000, seed=4).cuda()
for nw in (8, 12, 16):
    ms, _ = run(ctx8, q8, nw)
    print(f"B=8 Q=10000 nw={nw}: {ms*1e3:.1f} us")
t = time.time(); 
for _ in range(20): c = h.decode_latents(z[:1].contiguous(), use_graph=False)
torch.cuda.synchronize(); print("decode_latents B=1 eager ms", (time.time()-t)/20*1e3)
