"""AE legs for rocprofv3: encode (P = 10 000), latent stack and the streaming query decoder (Q = 1.2 M) at the batch given
(eager launches: RALD_GRAPH=0 would be read by the Python layer only; graphs are bypassed explicitly here)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rald_amd import bench_ae, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
m = bench_ae.build_ae(); h = m._handle()
pc = synth.point_cloud(B, 10000).cuda(); eps = synth.normal([B, 512, 32], 3).cuda()
q = synth.queries(1, 1200000).cuda()
for _ in range(reps):
    z = h.encode(pc, eps)[1]
    ctx = h.decode_latents(z, use_graph=False)
    ctx1 = h.decode_latents(z[:1].contiguous(), use_graph=False) if B > 1 else ctx
    h.decode_queries(ctx1, q)
torch.cuda.synchronize()
