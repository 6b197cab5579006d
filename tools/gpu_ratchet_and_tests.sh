# regenerate the accuracy ratchet with the library as built, copy it next to the truth fixtures (the GPU box's copy), run the -m gpu suite
set -o pipefail
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O
timeout -k 10 900 python tools/make_ratchet.py > $O/ratchet.log 2>&1 || { tail -5 $O/ratchet.log; exit 1; }
tail -3 $O/ratchet.log
cp $O/accuracy_ratchet.json tests/golden/accuracy_ratchet.json
bash tools/gpu_run_tests.sh
