#!/usr/bin/env bash
# timing-only experiments for the sb2st kernel (results invalid when BSP_SB2ST_DBG != 0)
for cfg in "0 128" "1 128" "2 128" "3 128" "0 16" "3 16" "0 32" "0 64"; do
  set -- $cfg
  BSP_SB2ST_DBG=$1 timeout -k 10 200 python bench.py --channels $2 --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('dbg=$1 channels=$2 sb2st_ms=%.1f sy2sb_ms=%.1f bisect_ms=%.1f' % (d['stage_ms_per_step_rank0']['sb2st'], d['stage_ms_per_step_rank0']['sy2sb'], d['stage_ms_per_step_rank0']['bisect']))
" 
done
