#!/bin/bash
# round-3 verdict, item 1: go / no-go timing of the fused update + symm sweep (upper bound: 16 K-steps per tile, no symm launch)
set -o pipefail
O=gpurun_out/r4f; mkdir -p $O
for ch in 128 16; do
  timeout -k 10 200 python tools/stage_times.py --channels $ch route=1 >> $O/fused_probe.txt 2>> $O/err.txt || { tail -5 $O/err.txt; exit 1; }
  timeout -k 10 200 python tools/stage_times.py --channels $ch route=1 fused_probe=1 >> $O/fused_probe.txt 2>> $O/err.txt || { tail -5 $O/err.txt; exit 1; }
done
cat $O/fused_probe.txt
