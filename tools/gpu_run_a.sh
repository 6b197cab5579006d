# round-2 GPU session A: the whole -m gpu suite, the truth report, the micro-benchmarks, one bench line
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out; mkdir -p $O; rm -f $O/stage_metrics.txt
echo "== pytest" ; timeout -k 10 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_a.log 2>&1; echo "pytest exit $?"; tail -15 $O/pytest_a.log
echo "== truth report"; timeout -k 10 300 python tools/truth_report.py > $O/truth_report.txt 2> $O/truth_report.err; echo "exit $?"
echo "== microbench"; (make -C tools/microbench > $O/microbench_build.log 2>&1 && cd tools/microbench && for w in 1 2 4; do echo "# mfma_f64_peak, $w workgroup(s) of 4 waves per CU"; timeout -k 5 120 ./mfma_f64_peak $w; done; echo "# gemm_core"; timeout -k 5 120 ./gemm_core; echo "# tile_bw"; timeout -k 5 120 ./tile_bw) > $O/microbench.txt 2>&1; echo "exit $?"
echo "== bench"; timeout -k 10 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_a.json 2> $O/bench_a.err; echo "exit $?"; cut -c1-600 $O/bench_a.json
