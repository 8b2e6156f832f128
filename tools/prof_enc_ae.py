"""AE encode only, for rocprofv3: P = 10 000 points at the batch given."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rald_amd import bench_ae, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
m = bench_ae.build_ae(); h = m._handle()
pc = synth.point_cloud(B, 10000).cuda(); eps = synth.normal([B, 512, 32], 3).cuda()
for _ in range(reps):
    z = h.encode(pc, eps)[1]
torch.cuda.synchronize()
