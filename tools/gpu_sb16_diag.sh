# per-wave cycle stamps of the band-16 chase (BSP_SB2ST_DIAG=1) for both layouts, then the A/B timing
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/ab
for r in 0 1; do
  echo "== BSP_SB16_ROWS=$r"
  BSP_SB16_ROWS=$r BSP_SB2ST_DIAG=1 timeout -k 10 300 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --channels ${1:-128} 2>&1 >/dev/null | grep "sb16st wave" | sort | uniq -c | sort -rn | head -12
done
