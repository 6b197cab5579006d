# end of a session: the -m gpu suite, smoke, the small-batch table, then gpu_profiles.sh (default route) -- everything profiles/r04_* is cut from
set -o pipefail
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/final; mkdir -p $O; rm -f gpurun_out/stage_metrics.txt $O/small_batch.txt
timeout -k 10 1000 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest.log 2>&1; rc=$?
echo "pytest exit $rc"; tail -6 $O/pytest.log | cut -c1-300
[ $rc -ge 124 ] && exit $rc
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -3 $O/smoke.log | cut -c1-200
for ch in 128 64 32 16; do
  timeout -k 10 200 python tools/stage_times.py --channels $ch --reps 5 >> $O/small_batch.txt 2>> $O/err.txt || { tail -5 $O/err.txt; exit 1; }
done
for ch in 128 16; do
  timeout -k 10 200 python tools/stage_times.py --channels $ch --reps 3 route=1 >> $O/small_batch.txt 2>> $O/err.txt || { tail -5 $O/err.txt; exit 1; }
done
cat $O/small_batch.txt
bash tools/gpu_profiles.sh
