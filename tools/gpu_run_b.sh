set -o pipefail
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O
for v in "BSP_BISECT_TAIL=1" "BSP_BISECT_TAIL=0" "BSP_BISECT_TAIL=1 BSP_BISECT_EPT=4" "BSP_BISECT_TAIL=0 BSP_BISECT_EPT=4"; do
  echo "== $v"; env $v timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['stage_ms_per_step_rank0'])"
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_b -o s -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/prof_b.log 2>&1
cd $GRAFT_REPO_ROOT; find $O/prof_b -name '*kernel_trace.csv' -delete; find $O/prof_b -name '*agent_info.csv' -delete
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/prof_b/**/*kernel_stats.csv',recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print("%-70s calls %5s avg %10.3f ms total %9.1f ms %5s%%" % (r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e6, float(r['TotalDurationNs'])/1e6, r['Percentage']))
PY
