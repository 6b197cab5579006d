#!/usr/bin/env python3
"""LDS and issue figures per kernel from rocprofv3 counter passes of `python bench.py --steps 1 --warmup 0 --no-cpu-baseline`
(tools/gpu_lds_counters.sh; --pmc with --kernel-trace only, separate passes):
  lds_array_busy  = SQ_LDS_IDX_ACTIVE / CU-cycles        (cycles the LDS array works, per CU; SQ_LDS_BANK_CONFLICT = the part of them
                                                          that bank conflicts add -- MI355X_MICROARCH.md, LDS)
  valu_busy       = SQ_ACTIVE_INST_VALU / SIMD-cycles
  lds per wave    = SQ_INSTS_LDS / waves, wait share = SQ_WAIT_INST_LDS / SQ_WAVE_CYCLES
CU-cycles = GRBM_GUI_ACTIVE / 8 XCDs * 256 CUs (GRBM_GUI_ACTIVE is reported as the sum over the XCDs).
usage: tools/lds_util.py <counter dir> [<counter dir> ...] <out.json>"""
import csv, glob, json, os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import kernel_sources_sha
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for d in sys.argv[1:-1]:
    f = glob.glob(d + "/*counter_collection.csv")[0]
    seen = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        agg[k][r["Counter_Name"] + "@" + os.path.basename(d)] += float(r["Counter_Value"])
        agg[k][r["Counter_Name"]] = agg[k][r["Counter_Name"] + "@" + os.path.basename(d)]
out = {"source": "rocprofv3 --kernel-trace --pmc <counters> -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-timing (tools/gpu_lds_counters.sh)",
       "csrc_sha16": kernel_sources_sha(), "kernels": {}}
for k, v in sorted(agg.items(), key=lambda x: -x[1].get("SQ_LDS_IDX_ACTIVE", 0)):
    gui = v.get("GRBM_GUI_ACTIVE", 0.0)
    if gui <= 0 or v.get("SQ_LDS_IDX_ACTIVE", 0) <= 0:
        continue
    cu_cycles, simd_cycles = gui / 8 * 256, gui / 8 * 1024
    e = {"lds_array_busy": v["SQ_LDS_IDX_ACTIVE"] / cu_cycles,
         "lds_bank_conflict_share_of_array_cycles": v.get("SQ_LDS_BANK_CONFLICT", 0.0) / v["SQ_LDS_IDX_ACTIVE"],
         "valu_busy": v.get("SQ_ACTIVE_INST_VALU", 0.0) / simd_cycles if "SQ_ACTIVE_INST_VALU" in v else None,
         "lds_instructions": v.get("SQ_INSTS_LDS"), "valu_instructions": v.get("SQ_INSTS_VALU"),
         "wave_cycles_waiting_for_lds": (v["SQ_WAIT_INST_LDS"] / v["SQ_WAVE_CYCLES"]) if v.get("SQ_WAVE_CYCLES") and "SQ_WAIT_INST_LDS" in v else None,
         "raw": {c: x for c, x in v.items() if "@" not in c}}
    out["kernels"][k] = e
    print("%-44s LDS array busy %.2f (conflicts %.2f of it)  VALU busy %s  LDS wait share %s" % (
        k[:44], e["lds_array_busy"], e["lds_bank_conflict_share_of_array_cycles"],
        "%.2f" % e["valu_busy"] if e["valu_busy"] is not None else "-",
        "%.2f" % e["wave_cycles_waiting_for_lds"] if e["wave_cycles_waiting_for_lds"] is not None else "-"))
json.dump(out, open(sys.argv[-1], "w"), indent=1)
