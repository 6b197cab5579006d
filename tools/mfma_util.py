#!/usr/bin/env python3
"""MFMA pipe utilisation per kernel from one rocprofv3 counter pass (tools/mfma_util.sh):
util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs).  SQ_VALU_MFMA_BUSY_CYCLES sums the busy cycles
of all SIMDs (64 per v_mfma_f64_16x16x4: it equals 64 x the number of MFMAs the kernel's flops imply), GRBM_GUI_ACTIVE
is reported as the sum over the 8 XCDs.  Counter passes serialise the kernels: standalone figures.
usage: tools/mfma_util.py <counter dir> <out.json>"""
import csv, glob, json, os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import kernel_sources_sha
f = glob.glob(sys.argv[1] + "/*counter_collection.csv")[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        n[k] += 1
out = {"source": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline",
       "formula": "mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024)", "csrc_sha16": kernel_sources_sha(), "kernels": {}}
for k, v in sorted(agg.items(), key=lambda x: -x[1].get("SQ_VALU_MFMA_BUSY_CYCLES", 0)):
    mf, gui = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), v.get("GRBM_GUI_ACTIVE", 0.0)
    if mf <= 0:
        continue
    out["kernels"][k] = {"launches": n[k], "mfma_busy_cycles": mf, "gui_active_cycles_sum8": gui, "mfma_util": mf / (gui / 8 * 1024),
                         "implied_tflops_f64": mf / 64 * 2048 / (gui / 8 / 2.4e9) / 1e12}
    print("%-50s launches %4d  MFMA util %.3f  (%.1f TFLOP/s at 2.4 GHz)" % (k[:50], n[k], out["kernels"][k]["mfma_util"], out["kernels"][k]["implied_tflops_f64"]))
json.dump(out, open(sys.argv[2], "w"), indent=1)
