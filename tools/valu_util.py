#!/usr/bin/env python3
"""Vector-ALU occupancy per kernel from one rocprofv3 counter pass (tools/gpu_profiles.sh): SQ_ACTIVE_INST_VALU / SQ_BUSY_CYCLES
(share of the busy cycles in which a SIMD issues a vector instruction), SQ_INSTS_VALU per launch.  usage: tools/valu_util.py <dir>"""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/*counter_collection.csv")[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_INSTS_VALU":
        n[k] += 1
for k, v in sorted(agg.items(), key=lambda x: -x[1].get("SQ_BUSY_CYCLES", 0))[:8]:
    print("%-44s launches %5d  VALU insts %.3e  ACTIVE_INST_VALU / BUSY_CYCLES %.3f  WAVE_CYCLES %.3e" % (
        k[:44], n[k], v.get("SQ_INSTS_VALU", 0), v.get("SQ_ACTIVE_INST_VALU", 0) / max(v.get("SQ_BUSY_CYCLES", 1), 1), v.get("SQ_WAVE_CYCLES", 0)))
