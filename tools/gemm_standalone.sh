# standalone durations of the sy2sb kernels (one pipeline, no look-ahead, so nothing overlaps): sums over one step
cd /tmp && export TMPDIR=/tmp
export BSP_SY2SB_GROUPS=1 BSP_SY2SB_LOOKAHEAD=0
rm -rf /tmp/px; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/px -o t -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline > /tmp/px.log 2>&1
f=$(find /tmp/px -name '*kernel_stats.csv' | head -1)
python3 -c "
import csv
for r in csv.DictReader(open('$f')):
    if any(k in r['Name'] for k in ('gemm', 'syr2k', 'tsmm', 'panel_qr', 'form_T', 'splitk')): print('%-60s calls %4s total %7.1f ms' % (r['Name'][5:65], r['Calls'], float(r['TotalDurationNs'])/1e6))
"
tail -1 /tmp/px.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('rydberg', d['rydberg_max_rel_err_n<=8'])"
