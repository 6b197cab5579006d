#!/bin/bash
set -o pipefail
O=gpurun_out/r4m; mkdir -p $O; rm -f $O/*.txt $O/*.log
timeout -k 10 600 python -m pytest tests/test_gpu_stages.py tests/test_gpu_solve.py -x -q -k "crawford or band_route" > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log | cut -c1-300; exit 1; }
tail -1 $O/pytest.log
for sp in 50 33 25 12; do
  BSP_CW_SPLIT=$sp timeout -k 10 600 python tools/make_ratchet.py > $O/ratchet_$sp.log 2>&1; echo "split $sp: make_ratchet exit $?" >> $O/times.txt
  grep "OVER THE GATE" $O/ratchet_$sp.log | sed "s/ route 2.*OVER THE GATE/ OVER/" >> $O/times.txt
  timeout -k 10 200 python tools/stage_times.py --channels 128 --reps 5 cw_split=$sp >> $O/times.txt 2>> $O/err.txt || { tail -5 $O/err.txt; exit 1; }
done
cat $O/times.txt
