# round 3: full -m gpu suite, then throughput at 128/64/32/16 channels and the 16-channel panel timeline
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3b; rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" ; tail -5 $O/pytest.log
for c in 128 64 32 16; do
  timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --channels $c > $O/bench$c.json 2> $O/bench$c.err || echo "bench $c failed"
done
cd /tmp && export TMPDIR=/tmp
for c in 16 128; do
  rm -rf /tmp/kt$c
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt$c -o t -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --channels $c > $O/trace$c.log 2>&1
  python3 $R/tools/panel_timeline.py $(find /tmp/kt$c -name '*kernel_trace.csv' | sed -n 1p) > $O/panel_timeline_$c.txt
done
cd $R
python - <<'PY'
import json, os
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "r3b")
for c in (128, 64, 32, 16):
    try:
        d = json.load(open(os.path.join(O, "bench%d.json" % c)))
        print("channels %3d: %.2f/s %.1f ms/step" % (c, d["value"], d["ms_per_step"]), {k: round(v, 1) for k, v in d["stage_ms_per_step_rank0"].items()})
    except Exception as e:
        print(c, "no line", e)
PY
