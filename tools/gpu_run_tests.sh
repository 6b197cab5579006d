set -o pipefail
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O; rm -f $O/stage_metrics.txt
timeout -k 10 1000 python -m pytest tests -m gpu -q -p no:cacheprovider "$@" > $O/pytest.log 2>&1; rc=$?; echo "pytest exit $rc"; tail -25 $O/pytest.log; exit $rc
