#!/bin/bash
# round 4: the whole -m gpu suite, smoke, the default bench line (with the dense and full-V legs)
set -o pipefail
O=gpurun_out/r4d; mkdir -p $O; rm -f gpurun_out/stage_metrics.txt
timeout -k 10 1000 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest.log 2>&1; rc=$?
echo "pytest exit $rc"; tail -12 $O/pytest.log | cut -c1-300
[ $rc -ge 124 ] && exit $rc
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || { tail -5 $O/smoke.log; }
tail -3 $O/smoke.log
timeout -k 10 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err || { tail -20 $O/bench_default.err; exit 1; }
python - <<'PY'
import json
d=json.load(open('gpurun_out/r4d/bench_default.json'))
print(d['value'], d['ms_per_step'], d['stage_ms_per_step_rank0'])
print('roofline', d['roofline']['frac'], d['roofline']['dense_algorithm_ceiling'])
print('dense', d.get('dense_two_stage',{}).get('value'), d.get('dense_two_stage',{}).get('roofline',{}).get('frac'))
print('fullV', {k:v for k,v in d.get('full_V',{}).items() if k!='what'})
print('cpu', d.get('cpu_baseline'))
PY
