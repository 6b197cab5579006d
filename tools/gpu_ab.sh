# A/B timing of bench.py under environment switches: tools/gpu_ab.sh CHANNELS "ENV=.. ENV=.." "..." ; one line per setting
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/ab
ch=$1; shift
for cfg in "$@"; do
  for c in $ch; do
    env $cfg timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --channels $c > gpurun_out/ab/line.json 2> gpurun_out/ab/err.txt || { echo "FAILED: $cfg"; tail -3 gpurun_out/ab/err.txt; }
    python -c "
import json; d=json.load(open('gpurun_out/ab/line.json')); print('%-40s ch %3d: %.2f/s %.1f ms/step' % ('$cfg', $c, d['value'], d['ms_per_step']), {k: round(v, 1) for k, v in d['stage_ms_per_step_rank0'].items()})"
  done
done
