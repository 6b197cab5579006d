# L2 (TCC) counters of the kernels of one bench step: requests, hit rate, tag stalls -> gpurun_out/l2/l2_util.txt
# (the TCC_EA_* request counters by size do not collect on this pool's rocprofv3: empty passes)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/l2; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE --output-format csv -d $O/p1 -o a -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-timing > $O/p1.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_TAG_STALL_sum --output-format csv -d $O/p3 -o c -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-timing > $O/p3.log 2>&1
cd $R
python3 - <<'PY' > $O/l2_util.txt 2>&1
import csv, glob, collections, os
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/l2"
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for d in ("p1", "p3"):
    fs = glob.glob(O + "/" + d + "/*counter_collection.csv")
    if not fs:
        print("no counters in", d); continue
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in sorted(agg.items(), key=lambda x: -x[1].get("TCC_REQ_sum", 0))[:14]:
    req = v.get("TCC_REQ_sum", 0) or 1
    print("%-40s L2 requests %.3g  hit rate %.2f  tag-stall cycles per L2 channel / kernel cycles %.3f" % (
        k[:40], req, v.get("TCC_HIT_sum", 0) / max(v.get("TCC_HIT_sum", 0) + v.get("TCC_MISS_sum", 0), 1),
        v.get("TCC_TAG_STALL_sum", 0) / 128.0 / max(v.get("GRBM_GUI_ACTIVE", 1) / 8.0, 1)))
PY
find $O -name '*kernel_trace.csv' -delete; find $O -name '*agent_info.csv' -delete; find $O -name '*counter_collection.csv' -delete
cat $O/l2_util.txt
