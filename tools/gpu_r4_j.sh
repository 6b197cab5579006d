#!/bin/bash
set -o pipefail
O=gpurun_out/r4j; mkdir -p $O; rm -f $O/times.txt
timeout -k 10 300 python tools/dbg_items4.py || exit 1
for ch in 128 16; do for v in 1 2; do
  timeout -k 10 200 python tools/stage_times.py --channels $ch --reps 5 cw_items4=$v >> $O/times.txt 2>> $O/err.txt || { tail -5 $O/err.txt; exit 1; }
done; done
cat $O/times.txt
