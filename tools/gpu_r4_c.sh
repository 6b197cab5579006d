#!/bin/bash
# A/B of the reflector form in the band route's RQ loop: time and accuracy figures
set -o pipefail
O=gpurun_out/r4c; mkdir -p $O
for v in 0 1; do
  BSP_CW_ONEDIV=$v timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_$v.json 2> $O/bench_$v.err || { tail -5 $O/bench_$v.err; exit 1; }
  python -c "import json;d=json.load(open('$O/bench_$v.json'));print('onediv $v',round(d['value'],1),'/s',round(d['ms_per_step'],2),'ms',d['stage_ms_per_step_rank0'])"
  BSP_CW_ONEDIV=$v timeout -k 10 400 python tools/make_ratchet.py c3_1024_l31 c3_2048_l31 c4_4096_l127 c2_2048 sf2048 lin256 n128 bc1_2048 > $O/ratchet_$v.log 2>&1
  cut -c1-260 $O/ratchet_$v.log
done
