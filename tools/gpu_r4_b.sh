#!/bin/bash
# round 4: ratchet measurement of both routes (dense first: it seeds max_abs; then the default), bench lines, kernel statistics
set -o pipefail
O=gpurun_out/r4b; mkdir -p $O
timeout -k 10 500 python tools/make_ratchet.py --route 1 > $O/ratchet_dense.log 2>&1; rc=$?
[ $rc -ge 124 ] && { tail -5 $O/ratchet_dense.log; exit $rc; }
cp gpurun_out/accuracy_ratchet.json $O/accuracy_ratchet_after_dense.json
timeout -k 10 500 python tools/make_ratchet.py --base gpurun_out/accuracy_ratchet.json > $O/ratchet_default.log 2>&1; rc=$?
[ $rc -ge 124 ] && { tail -5 $O/ratchet_default.log; exit $rc; }
tail -4 $O/ratchet_dense.log; tail -4 $O/ratchet_default.log
for ch in 128 64 32 16; do
  timeout -k 10 300 python bench.py --steps 5 --warmup 2 --channels $ch --no-cpu-baseline > $O/bench_band_$ch.json 2> $O/bench_band_$ch.err || { tail -5 $O/bench_band_$ch.err; exit 1; }
  python -c "import json;d=json.load(open('$O/bench_band_$ch.json'));print($ch,'channels',round(d['value'],1),'/s',round(d['ms_per_step'],2),'ms',d['stage_ms_per_step_rank0'])"
done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-kernel-timing > $GRAFT_REPO_ROOT/$O/prof_bench.json 2> $GRAFT_REPO_ROOT/$O/prof.err || { tail -5 $GRAFT_REPO_ROOT/$O/prof.err; exit 1; }
cd $GRAFT_REPO_ROOT
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
head -12 $O/kernel_stats.csv
