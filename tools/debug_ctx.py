import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rald_amd import bench_ae, synth
m = bench_ae.build_ae(); h = m._handle()
for use_graph in (False, True):
    for seed in (1, 2, 1):
        z = synth.normal([1, 512, 32], seed).cuda()
        ctx = h.decode_latents(z, use_graph=use_graph)
        torch.cuda.synchronize()
        c = ctx.cpu().numpy()
        inv = np.frombuffer(c[512*128 + 512*4: 512*128 + 512*4 + 4].tobytes(), np.float32)[0]
        img = np.frombuffer(c[:512*128].tobytes(), np.float16).astype(np.float64)
        print("graph", use_graph, "seed", seed, "inv_scale", inv, "image absmax", np.abs(img).max(), "checksum", float(np.abs(img).sum()))
