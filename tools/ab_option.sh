#!/bin/bash
# A/B of one switch of the library on the bench workload (GPU): AB_NAME=BSP_VEC_EARLY AB_VALUES="1 0 1" bash tools/ab_option.sh
# prints eigensolves/s, ms per step and the stage times of every run
for v in ${AB_VALUES:-1 0}; do env ${AB_NAME:-BSP_VEC_EARLY}=$v timeout -k 10 200 python bench.py --steps 10 --warmup 3 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('${AB_NAME:-BSP_VEC_EARLY}=$v', round(j['value'],1), round(j['ms_per_step'],2), {k: round(x,2) for k,x in j['stage_ms_per_step_rank0'].items()})
"; done
