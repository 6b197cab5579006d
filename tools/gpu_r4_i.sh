#!/bin/bash
set -o pipefail
O=gpurun_out/r4i; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_stages.py -q -x -k "standard_form or crawford or not_positive" > $O/pytest.log 2>&1 || { tail -20 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
timeout -k 10 300 python tools/make_ratchet.py --route 1 c3_1024_l31 c5_1024_k11 sf2048 > $O/r1.log 2>&1; cut -c1-200 $O/r1.log
timeout -k 10 300 python tools/make_ratchet.py c3_1024_l31 c4_4096_l127 bc1 > $O/r2.log 2>&1; cut -c1-200 $O/r2.log
timeout -k 10 200 python tools/stage_times.py --channels 128 --reps 5
timeout -k 10 200 python tools/stage_times.py --channels 16 --reps 5
timeout -k 10 200 python tools/stage_times.py --channels 128 route=1
