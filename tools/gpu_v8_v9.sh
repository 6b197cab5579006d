# bulge chasing in one step (sb2st_version 8) against two steps (9): bench workload at 128/64/32/16 channels, and C2 / C3 / C5
cd "$GRAFT_REPO_ROOT"
for v in 8 9; do
  export BSP_SB2ST_VERSION=$v
  for c in 128 64 32 16; do
    timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --channels $c 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('version $v channels %3d: %.2f/s  %.1f ms/step ' % ($c, d['value'], d['ms_per_step']), {k: round(v,2) for k,v in d['stage_ms_per_step_rank0'].items()})"
  done
  timeout -k 10 300 python - <<PY
import sys, time, numpy as np
sys.path.insert(0, "tests")
from bspatom_amd import capi
from test_gpu_stages import input_from_case
for name, nl in (("c2_2048", 1), ("c3_2048_l31", 32), ("c5_8192", 1)):
    prob = capi.Problem(input_from_case(name))
    prob.solve(0, nl)
    E, info = prob.solve(0, nl)
    print("version $v %-12s n=%d channels=%d:" % (name, prob.nfun, nl), {k: round(v, 2) for k, v in prob.last_timing().items()})
    prob.close()
PY
done
