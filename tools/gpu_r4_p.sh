#!/bin/bash
set -o pipefail
O=gpurun_out/r4p; mkdir -p $O; rm -f $O/*.txt $O/*.log
timeout -k 10 900 python -m pytest tests/test_gpu_solve.py tests/test_gpu_stages.py -x -q -k "bisect or spectra_vs_reference or batch_size or band_route or variants_at_scale or c4_all or c3_at_full" > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log | cut -c1-300; exit 1; }
tail -1 $O/pytest.log
for c in 128 16; do for v in bisect_secant=0 bisect_secant=1; do
  timeout -k 10 200 python tools/stage_times.py --channels $c --reps 5 $v >> $O/times.txt 2>> $O/err.txt || { tail -5 $O/err.txt; exit 1; }
done; done
cat $O/times.txt
