# A/B timing: every argument is one environment assignment list ("" = defaults); prints eigensolves/s and stage times
set -o pipefail
cd "$GRAFT_REPO_ROOT"
for v in "$@"; do
  echo "== [$v]"; env $v timeout -k 10 300 python bench.py --steps ${STEPS:-3} --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f/s  %.1f ms/step ' % (d['value'], d['ms_per_step']), {k: round(v,2) for k,v in d['stage_ms_per_step_rank0'].items()})"
done
