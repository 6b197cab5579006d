# the -m gpu suite against the COMMITTED ratchet first; then a measurement that can only tighten it (gpurun_out/accuracy_ratchet.json,
# copied over tests/golden/ by hand after reading the diff -- make_ratchet.py exits 1 if a figure is over its committed gate)
set -o pipefail
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O
bash tools/gpu_run_tests.sh || exit 1
timeout -k 10 900 python tools/make_ratchet.py > $O/ratchet.log 2>&1; rc=$?
tail -5 $O/ratchet.log
exit $rc
