"""Scratch: AE encode + decode_latents at one batch size, for rocprofv3 --kernel-trace --stats."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rald_amd import bench_ae, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
os.environ.setdefault("RALD_GRAPH", "0")
m = bench_ae.build_ae(); h = m._handle()
pc = synth.point_cloud(B, 10000).cuda(); eps = synth.normal([B, 512, 32], 3).cuda()
for _ in range(12):
    z = h.encode(pc, eps)[1]
    ctx = h.decode_latents(z)
torch.cuda.synchronize()
