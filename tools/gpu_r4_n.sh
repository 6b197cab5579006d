#!/bin/bash
set -o pipefail
O=gpurun_out/r4n; mkdir -p $O; rm -f $O/*.txt $O/*.log
timeout -k 10 600 python -m pytest tests/test_gpu_stages.py tests/test_gpu_solve.py -x -q -k "crawford or band_route or spectra_vs_reference" > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log | cut -c1-300; exit 1; }
tail -1 $O/pytest.log
for ch in 128 64 16 1; do for v in cw_graph=0 cw_graph=1; do
  timeout -k 10 200 python tools/stage_times.py --channels $ch --reps 5 $v >> $O/times.txt 2>> $O/err.txt || { tail -5 $O/err.txt; exit 1; }
done; done
cat $O/times.txt
