#!/bin/bash
# the four-items-per-wave chase kernel: stage test, a few solves, stage times of the variants
set -o pipefail
O=gpurun_out/r4g; mkdir -p $O; rm -f $O/times.txt
timeout -k 10 600 python -m pytest tests/test_gpu_stages.py -x -q -k "crawford" > $O/pytest_crawford.log 2>&1 || { tail -30 $O/pytest_crawford.log; exit 1; }
tail -2 $O/pytest_crawford.log
timeout -k 10 900 python -m pytest tests/test_gpu_solve.py -x -q -k "(c3_1024 or c2_2048 or rydberg or band_route or c4_channels or not_positive) and not dense" > $O/pytest_solve.log 2>&1 || { tail -40 $O/pytest_solve.log | cut -c1-300; exit 1; }
tail -2 $O/pytest_solve.log
for ch in 128 16; do for v in "cw_items4=0" "cw_items4=1 cw_nw=4" "cw_items4=1 cw_nw=1"; do
  timeout -k 10 200 python tools/stage_times.py --channels $ch $v >> $O/times.txt 2>> $O/err.txt || { tail -5 $O/err.txt; exit 1; }
done; done
cat $O/times.txt
