# round 3: the files of profiles/ that are not counters: throughput by batch size, truth report, panel timelines, 'V' timing
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3f; rm -rf $O; mkdir -p $O
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
  echo "# the same with the round-2 panel route (BSP_PANEL_QR=2: one workgroup per channel for every panel) and its bisection shape (BSP_BISECT_EPT=5 at 128, 4 below)"
  for c in 128 16; do
    e=5; [ $c -lt 128 ] && e=4
    BSP_PANEL_QR=2 BSP_BISECT_EPT=$e timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --channels $c > $O/o$c.json 2> $O/o$c.err
    python -c "
import json; d=json.load(open('$O/o$c.json')); s=d['stage_ms_per_step_rank0']
print('channels %3d: %7.2f eigensolves/s  %6.1f ms/step | assemble %.1f  cholesky+standard form %.1f  sy2sb %.1f  bulge chasing %.1f  bisection %.1f' % ($c, d['value'], d['ms_per_step'], s['assemble'], s['chol_std'], s['sy2sb'], s['sb2st'], s['bisect']))"
  done
} > $O/small_batch_table.txt
cat $O/small_batch_table.txt
timeout -k 10 600 python tools/truth_report.py c3_1024_l31 c5_1024_k11 lin1024 c2_2048 c4_4096 c5_8192 > $O/truth_report.txt 2>&1; tail -3 $O/truth_report.txt | cut -c1-200
bash tools/gpu_trace.sh 16 128 && cp gpurun_out/trace/panel_timeline_16.txt gpurun_out/trace/panel_timeline_128.txt $O/
