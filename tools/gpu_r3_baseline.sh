# round 3, first session: accuracy ratchet of the round-2 kernels, bench line with live kernel timing, small-batch regime + its kernel trace
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3a; rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 900 python tools/make_ratchet.py > $O/ratchet.log 2>&1 && cp gpurun_out/accuracy_ratchet.json $O/
timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench128.json 2> $O/bench128.err
for c in 64 32 16; do
  timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --channels $c > $O/bench$c.json 2> $O/bench$c.err
done
cd /tmp && export TMPDIR=/tmp
for c in 16 32; do
  rm -rf /tmp/kt$c
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt$c -o t -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --channels $c > $O/trace$c.log 2>&1
  python3 $R/tools/panel_timeline.py $(find /tmp/kt$c -name '*kernel_trace.csv' | sed -n 1p) > $O/panel_timeline_$c.txt
done
cd $R
python - <<'PY'
import json, glob, os
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "r3a")
for c in (128, 64, 32, 16):
    d = json.load(open(os.path.join(O, "bench%d.json" % c)))
    print("channels %3d: %.2f/s %.1f ms/step" % (c, d["value"], d["ms_per_step"]), {k: round(v, 1) for k, v in d["stage_ms_per_step_rank0"].items()})
PY
