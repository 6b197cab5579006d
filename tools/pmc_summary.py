#!/usr/bin/env python3
"""Turn two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, --kernel-trace only) of
`python bench.py --steps 1 --warmup 0 --no-cpu-baseline` into profiles/<tag>_pmc_summary.json.

Units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): the counters are in
KiB; on gfx950 FETCH_SIZE reports half of the bytes of a coalesced streaming read, so it is doubled;
WRITE_SIZE is exact.  usage: tools/pmc_summary.py <dir_FETCH> <dir_WRITE> <out.json> [channels nfun]"""
import csv, glob, json, os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import kernel_sources_sha

def per_kernel(d, name):
    f = glob.glob(d + "/*counter_collection.csv")[0]
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == name:
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
            agg[k][0] += 1; agg[k][1] += float(r["Counter_Value"])
    return agg

fd, wd, out = sys.argv[1:4]
F, W = per_kernel(fd, "FETCH_SIZE"), per_kernel(wd, "WRITE_SIZE")
res = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline",
       "units": "KiB raw; traffic_bytes_per_launch = (2*FETCH_SIZE + WRITE_SIZE) * 1024 / launches (gfx950 FETCH_SIZE x2 correction)",
       "workload": {"channels": int(sys.argv[4]) if len(sys.argv) > 4 else 128, "nfun": int(sys.argv[5]) if len(sys.argv) > 5 else 4096},
       "csrc_sha16": kernel_sources_sha(),      # the kernel sources the counters were taken with (bench.py flags a mismatch as stale)
       "kernels": {}}
for k in sorted(set(F) | set(W), key=lambda k: -(2 * F.get(k, [0, 0])[1] + W.get(k, [0, 0])[1])):
    n = max(F.get(k, [0, 0])[0], W.get(k, [0, 0])[0])
    f, w = F.get(k, [0, 0.0])[1], W.get(k, [0, 0.0])[1]
    res["kernels"][k] = {"launches": n, "fetch_kib_raw": f, "write_kib": w,
                         "traffic_bytes_per_launch": (2 * f + w) * 1024 / max(n, 1)}
json.dump(res, open(out, "w"), indent=1)
for k, v in list(res["kernels"].items())[:6]:
    print("%-50s launches %4d  traffic/launch %.3f GB" % (k[:50], v["launches"], v["traffic_bytes_per_launch"] / 1e9))
