R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c; rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_stages.py -x -q -k "panel_factorisation or sy2sb" > $O/pytest_panel.log 2>&1; echo "panel tests rc=$?"; tail -3 $O/pytest_panel.log
BSP_TSQR_REGCAP=1 timeout -k 10 600 python -m pytest tests/test_gpu_stages.py -x -q -k "panel_factorisation" > $O/pytest_panel_cap.log 2>&1; echo "panel tests (regcap) rc=$?"; tail -3 $O/pytest_panel_cap.log
for cap in 0 1; do for c in 128 16; do
  BSP_TSQR_REGCAP=$cap timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --channels $c > $O/bench${c}_cap$cap.json 2> $O/bench${c}_cap$cap.err || echo "bench $c failed"
  python -c "
import json; d=json.load(open('$O/bench${c}_cap$cap.json')); print('cap $cap channels %3d: %.2f/s %.1f ms/step' % ($c, d['value'], d['ms_per_step']), {k: round(v, 1) for k, v in d['stage_ms_per_step_rank0'].items()})"
done; done
bash tools/gpu_trace.sh 16
