#!/bin/bash
set -o pipefail
O=gpurun_out/r4o; mkdir -p $O; rm -f $O/*.txt $O/*.log
for v in cw_onediv=0 cw_onediv=1; do
  timeout -k 10 200 python tools/stage_times.py --channels 128 --reps 5 $v >> $O/times.txt 2>> $O/err.txt || { tail -5 $O/err.txt; exit 1; }
done
BSP_CW_ONEDIV=1 timeout -k 10 600 python tools/make_ratchet.py > $O/ratchet_onediv.log 2>&1; echo "onediv: make_ratchet exit $?" >> $O/times.txt
grep "OVER THE GATE" $O/ratchet_onediv.log | sed "s/ route 2.*OVER THE GATE/ OVER/" >> $O/times.txt
cat $O/times.txt
