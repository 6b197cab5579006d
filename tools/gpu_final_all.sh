# end of a session: the -m gpu suite, then the raw material of profiles/ (kernel stats, counter passes, MFMA utilisation, micro-benchmarks)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
bash tools/gpu_run_tests.sh || exit 1
bash tools/gpu_profiles.sh > gpurun_out/prof_run.log 2>&1; tail -3 gpurun_out/prof_run.log | cut -c1-300
