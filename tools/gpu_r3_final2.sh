# round 3, second session: throughput by batch size, truth report, the LDS / barrier micro-benchmarks, stamps of sb16r_kernel
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3f2; rm -rf $O; mkdir -p $O
cd $R
{
  echo "# eigensolves/s and stage times (ms per step, HIP events) by channels per GPU: what each GPU of BASELINE configs[3] sees at N = 1, 2, 4, 8"
  echo "# python bench.py --steps 4 --warmup 1 --no-cpu-baseline --channels C   (n = 4096, k = 9, rb = 800; one MI355X)"
  for c in 128 64 32 16; do
    timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --channels $c > $O/b$c.json 2> $O/b$c.err
    python -c "
import json; d=json.load(open('$O/b$c.json')); s=d['stage_ms_per_step_rank0']
print('channels %3d: %7.2f eigensolves/s  %6.1f ms/step | assemble %.1f  cholesky+standard form %.1f  sy2sb %.1f  bulge chasing %.1f  bisection %.1f' % ($c, d['value'], d['ms_per_step'], s['assemble'], s['chol_std'], s['sy2sb'], s['sb2st'], s['bisect']))"
  done
  echo "# the same with the first layout of the band-16 chase (BSP_SB16_ROWS=0: sb16st_kernel, the state of the round's first session)"
  for c in 128 16; do
    BSP_SB16_ROWS=0 timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --channels $c > $O/o$c.json 2> $O/o$c.err
    python -c "
import json; d=json.load(open('$O/o$c.json')); s=d['stage_ms_per_step_rank0']
print('channels %3d: %7.2f eigensolves/s  %6.1f ms/step | assemble %.1f  cholesky+standard form %.1f  sy2sb %.1f  bulge chasing %.1f  bisection %.1f' % ($c, d['value'], d['ms_per_step'], s['assemble'], s['chol_std'], s['sy2sb'], s['sb2st'], s['bisect']))"
  done
} > $O/small_batch_table.txt
cat $O/small_batch_table.txt
timeout -k 10 600 python tools/truth_report.py c3_1024_l31 c5_1024_k11 lin1024 c2_2048 c4_4096 c5_8192 > $O/truth_report.txt 2>&1; tail -3 $O/truth_report.txt | cut -c1-200
(make -C tools/microbench > /dev/null 2>&1; cd tools/microbench && echo "# barrier_lds" && ./barrier_lds && echo "# lds_rate" && ./lds_rate) > $O/microbench2.txt 2>&1
bash tools/gpu_sb16_diag.sh 128 > $O/sb16_stamps.txt 2>&1
bash tools/gpu_v8_v9.sh > $O/v8_v9.txt 2>&1; tail -12 $O/v8_v9.txt | cut -c1-200
python3 bench.py > $O/bench_line.json 2> $O/bench.err; tail -c 600 $O/bench_line.json
