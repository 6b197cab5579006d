# single-GPU throughput at 128/64/32/16 channels (what each GPU sees at N = 1/2/4/8 of BASELINE configs[3])
cd "$GRAFT_REPO_ROOT"
for c in 128 64 32 16; do
  timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --channels $c 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('channels %3d: %.2f/s  %.1f ms/step ' % ($c, d['value'], d['ms_per_step']), {k: round(v,2) for k,v in d['stage_ms_per_step_rank0'].items()})"
done
