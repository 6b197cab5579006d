# kernel-trace stats of one bench run: top kernels with average durations
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ks && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -o s -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /tmp/ks.log 2>&1
f=$(find /tmp/ks -name '*kernel_stats.csv' | sed -n 1p)
python3 - "$f" "$@" <<'PY'
import csv, sys
pat = sys.argv[2:] or [""]
for r in csv.DictReader(open(sys.argv[1])):
    if any(p in r["Name"] for p in pat):
        print("%-64s calls %5s avg %9.3f ms  min %9.3f  max %9.3f" % (r["Name"][:64], r["Calls"], float(r["AverageNs"]) / 1e6, float(r["MinNs"]) / 1e6, float(r["MaxNs"]) / 1e6))
PY
