#!/bin/bash
# round 4: quick contact of the band route with the GPU: stage test, a few solve cases, a bench line, kernel statistics
set -o pipefail
O=gpurun_out/r4a; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_stages.py -x -q -k "crawford" > $O/pytest_crawford.log 2>&1 || { tail -30 $O/pytest_crawford.log; exit 1; }
tail -2 $O/pytest_crawford.log
timeout -k 10 900 python -m pytest tests/test_gpu_solve.py -x -q -k "c3_1024 or c2_2048 or rydberg or band_route or c4_channels" > $O/pytest_solve.log 2>&1 || { tail -40 $O/pytest_solve.log; exit 1; }
tail -2 $O/pytest_solve.log
for ch in 128 16; do
timeout -k 10 600 python bench.py --steps 5 --warmup 2 --channels $ch --no-cpu-baseline > $O/bench_band_$ch.json 2> $O/bench_band.err || { tail -20 $O/bench_band.err; exit 1; }
python -c "import json;d=json.load(open('$O/bench_band_$ch.json'));print($ch,'channels',round(d['value'],1),'/s',round(d['ms_per_step'],2),'ms',d['stage_ms_per_step_rank0'])"
done
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats -o s -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-kernel-timing > $R/$O/stats.log 2>&1 || { tail -5 $R/$O/stats.log; exit 1; }
cd $R
cp $(find $O/stats -name '*kernel_stats.csv' | head -1) $O/kernel_stats.csv
find $O/stats -name '*kernel_trace.csv' -delete; find $O/stats -name '*agent_info.csv' -delete
head -8 $O/kernel_stats.csv | cut -c1-200
