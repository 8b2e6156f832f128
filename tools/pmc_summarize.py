"""Per-kernel summary of a rocprofv3 --pmc counter_collection.csv + kernel_trace.csv pair: median counters per launch,
MFMA-busy fraction and effective clock.  usage: python tools/pmc_summarize.py <dir> > profiles/<name>.csv"""
import csv, glob, os, sys, statistics as st
from collections import defaultdict
d = sys.argv[1]
cc = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
kt = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
dur = {}
for f in kt:
    for r in csv.DictReader(open(f)):
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
rows = defaultdict(lambda: defaultdict(dict))     # kernel -> dispatch -> counter -> value
for f in cc:
    for r in csv.DictReader(open(f)):
        rows[r["Kernel_Name"]][r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
def short(n):
    n = n.replace("rald::", "")
    return n if len(n) < 110 else n[:107] + "..."
print("kernel,launches,median_us,MFMA_BUSY_CYCLES,SQ_BUSY_CYCLES,SQ_WAVE_CYCLES,GRBM_GUI_ACTIVE,eff_clock_GHz,mfma_busy_frac_of_cu_cycles")
out = []
for k, disp in rows.items():
    ids = [i for i in disp if i in dur]
    if not ids:
        continue
    med = lambda c: st.median([disp[i].get(c, 0.0) for i in ids])
    us = st.median([dur[i] for i in ids]) / 1e3
    mf, sb, wc, gui = med("SQ_VALU_MFMA_BUSY_CYCLES"), med("SQ_BUSY_CYCLES"), med("SQ_WAVE_CYCLES"), med("GRBM_GUI_ACTIVE")
    clk = gui / 8.0 / (us * 1e3) if us > 0 else 0.0                  # GRBM_GUI_ACTIVE is summed over the 8 XCDs (MI355X_MICROARCH.md DVFS)
    # SQ_VALU_MFMA_BUSY_CYCLES: cycles a SIMD's matrix pipe is busy, summed over all SIMDs of the chip (256 CUs x 4); the kernel had
    # gui/8 cycles of wall time, so the busy fraction of all matrix pipes is mf / (1024 * gui / 8)
    frac = mf / (1024.0 * gui / 8.0) if gui > 0 else 0.0
    out.append((us * len(ids), f'"{short(k)}",{len(ids)},{us:.2f},{mf:.0f},{sb:.0f},{wc:.0f},{gui:.0f},{clk:.3f},{frac:.3f}'))
for _, line in sorted(out, reverse=True):
    print(line)
