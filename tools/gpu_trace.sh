# kernel trace of bench.py at the channel counts given ($@, default 16): panel timeline per count in gpurun_out/trace/
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/trace; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in ${@:-16}; do
  rm -rf /tmp/kt$c
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt$c -o t -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --channels $c > $O/trace$c.log 2>&1
  python3 $R/tools/panel_timeline.py $(find /tmp/kt$c -name '*kernel_trace.csv' | sed -n 1p) > $O/panel_timeline_$c.txt
done
