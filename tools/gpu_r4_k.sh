#!/bin/bash
set -o pipefail
O=gpurun_out/r4k; mkdir -p $O; rm -f $O/times.txt
for pad in 0 12 30 70; do
  timeout -k 10 200 python tools/stage_times.py --channels 128 --reps 3 cw_ldspad=$pad >> $O/times.txt 2>> $O/err.txt || { tail -5 $O/err.txt; exit 1; }
done
cat $O/times.txt
