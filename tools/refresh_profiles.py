"""Copy the round's evidence from gpurun_out/ (tools/r3_final_a.sh, r3_final_b.sh) into profiles/ and rebuild the in-situ HBM traffic entries of
profiles/traffic.json from the FETCH_SIZE / WRITE_SIZE passes (KB -> bytes; FETCH_SIZE doubled: the gfx950 correction of MI355X_MICROARCH.md's
HBM section for wide loads; WRITE_SIZE exact; median per launch).  The three uses of gemm_resid_ln per block (to_out, folded cross-attention
output, ff.net.2) are told apart by launch order.  usage: python tools/refresh_profiles.py [tag, default r03]"""
import csv, json, os, shutil, statistics, sys
from collections import defaultdict
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import bench
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
G, P = os.path.join(R, "gpurun_out"), os.path.join(R, "profiles")
copies = {
    "r3_bench_B128.json": f"{tag}_bench_B128.json", "r3_bench_B64.json": f"{tag}_bench_B64.json",
    "r3_bench_B128_under_rocprof.json": f"{tag}_bench_B128_under_rocprof.json", "r3_bench_B64_under_rocprof.json": f"{tag}_bench_B64_under_rocprof.json",
    "r3_prof_bench128/bench_kernel_stats.csv": f"{tag}_bench_B128_kernel_stats.csv", "r3_prof_bench64/bench_kernel_stats.csv": f"{tag}_bench_B64_kernel_stats.csv",
    "r3_prof_b1/b1_kernel_stats.csv": f"{tag}_nfe_B1_kernel_stats.csv", "r3_prof_train_final/train_kernel_stats.csv": f"{tag}_train_step_B8_kernel_stats.csv",
    "r3_pmc_mfma_B1.csv": f"{tag}_pmc_mfma_B1.csv", "r3_pmc_mfma_B64.csv": f"{tag}_pmc_mfma_B64.csv", "r3_pmc_mfma_B128.csv": f"{tag}_pmc_mfma_B128.csv",
    "r3_pmc_nfe_B64_FETCH_SIZE/nfe_counter_collection.csv": f"pmc/{tag}_nfe_B64_FETCH_SIZE.csv",
    "r3_pmc_nfe_B64_WRITE_SIZE/nfe_counter_collection.csv": f"pmc/{tag}_nfe_B64_WRITE_SIZE.csv",
}
for src, dst in copies.items():
    s = os.path.join(G, src)
    if os.path.exists(s):
        shutil.copyfile(s, os.path.join(P, dst))
    else:
        print("missing", src)

def per_kernel(B, counter):
    rows = defaultdict(list)
    with open(os.path.join(G, f"r3_pmc_nfe_B{B}_{counter}", "nfe_counter_collection.csv")) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] == counter:
                rows[r["Kernel_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"]) * 1024.0))
    return rows

tj = json.load(open(os.path.join(P, "traffic.json")))
insitu, table = {"_note": "fetch (x2 corrected) + write bytes per launch, median over eight NFEs; gemm_resid_ln's three uses per block told apart by launch order"}, {}
for B in (64, 128):
    fe, wr = per_kernel(B, "FETCH_SIZE"), per_kernel(B, "WRITE_SIZE")
    for k in fe:
        f = statistics.median(v for _, v in fe[k]) * 2.0
        w = statistics.median(v for _, v in wr.get(k, [(0, 0.0)]))
        short = k.replace("rald::", "")[:90]
        table[f"B{B} {short}"] = {"launches": len(fe[k]), "fetch_MB_per_launch_x2_corrected": round(f / 1e6, 1), "write_MB_per_launch": round(w / 1e6, 1)}
        if "gemm_nt_glds_kernel<256, 256, 4, 2, 2, 3>" in k:
            tj[f"ff1_geglu_gemm_B{B}"] = f + w
        if "gemm_resid_ln_kernel<128" in k:
            fs, ws = sorted(fe[k]), sorted(wr[k])
            for i, name in enumerate(("gemm_resid_ln_K512_attn1", "gemm_resid_ln_K512_attn2", "gemm_resid_ln_K2048_ff2")):
                per_nfe = len(fs) // 8                            # 71 = 24 blocks x 3 - the last block's ff.net.2 (no LayerNorm follows it)
                fsel = [v for j, (_, v) in enumerate(fs) if (j % per_nfe) % 3 == i]
                wsel = [v for j, (_, v) in enumerate(ws) if (j % per_nfe) % 3 == i]
                insitu[f"{name}_B{B}"] = statistics.median(fsel) * 2.0 + statistics.median(wsel)
tj["in_situ_bytes_per_launch"] = insitu
tj[f"{tag}_in_situ_MB_per_launch"] = table
tj["_source_sha"] = bench.source_fingerprint()
json.dump(tj, open(os.path.join(P, "traffic.json"), "w"), indent=1)
print("source sha", tj["_source_sha"]); print({k: round(v / 1e6, 1) for k, v in insitu.items() if not k.startswith("_")})
print("ff1 B64 / B128 MB:", round(tj["ff1_geglu_gemm_B64"] / 1e6, 1), round(tj["ff1_geglu_gemm_B128"] / 1e6, 1))
