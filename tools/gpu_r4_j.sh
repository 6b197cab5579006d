#!/bin/bash
set -o pipefail
O=gpurun_out/r4j; mkdir -p $O; rm -f $O/times.txt
timeout -k 10 300 python -m pytest tests/test_gpu_stages.py -q -x -k "crawford" > $O/pytest.log 2>&1 || { tail -20 $O/pytest.log; exit 1; }
tail -1 $O/pytest.log
for ch in 128 16; do
  timeout -k 10 200 python tools/stage_times.py --channels $ch --reps 5 >> $O/times.txt 2>> $O/err.txt || { tail -5 $O/err.txt; exit 1; }
done
cat $O/times.txt
