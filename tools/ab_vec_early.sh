#!/bin/bash
# A/B of BSP_VEC_EARLY (the consumed eigenvector's eigenvalue from the pencil's inertia, csrc/bandsect.hip) on the bench workload (GPU)
for v in ${AB_VALUES:-1 0 1}; do BSP_VEC_EARLY=$v timeout -k 10 200 python bench.py --steps 10 --warmup 3 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('vec_early=$v', round(j['value'],1), round(j['ms_per_step'],2), {k: round(x,2) for k,x in j['stage_ms_per_step_rank0'].items()})
"; done
