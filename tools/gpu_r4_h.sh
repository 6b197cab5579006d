#!/bin/bash
# final state of the session: whole suite, small-batch table, profiles of the default route
set -o pipefail
O=gpurun_out/r4h; mkdir -p $O; rm -f gpurun_out/stage_metrics.txt $O/small_batch.txt
timeout -k 10 1000 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest.log 2>&1; rc=$?
echo "pytest exit $rc"; tail -6 $O/pytest.log | cut -c1-300
[ $rc -ge 124 ] && exit $rc
for ch in 128 64 32 16; do
  timeout -k 10 200 python tools/stage_times.py --channels $ch --reps 5 >> $O/small_batch.txt 2>> $O/err.txt || { tail -5 $O/err.txt; exit 1; }
done
cat $O/small_batch.txt
