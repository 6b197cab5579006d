# regenerate the raw material of profiles/: kernel stats, two PMC passes (HBM bytes), MFMA utilisation, bench lines, for the default
# (band) route; with ROUTE=1 for the dense route.  usage: bash tools/gpu_profiles.sh  (on the GPU box; output in gpurun_out/prof[_dense])
set -e
R=$GRAFT_REPO_ROOT; RT=${ROUTE:-0}; O=$R/gpurun_out/prof; [ "$RT" = 1 ] && O=$R/gpurun_out/prof_dense; rm -rf $O; mkdir -p $O
B="$R/bench.py --no-cpu-baseline --no-kernel-timing --no-dense-leg --no-full-v --route $RT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python3 $B --steps 4 --warmup 1 > $O/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o f -- python3 $B --steps 1 --warmup 0 > $O/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o w -- python3 $B --steps 1 --warmup 0 > $O/write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/mfma -o m -- python3 $B --steps 1 --warmup 0 > $O/mfma.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/valu -o v -- python3 $B --steps 1 --warmup 0 > $O/valu.log 2>&1 || true
cd $R
python3 tools/pmc_summary.py $O/fetch $O/write $O/pmc_summary.json 128 4096 > $O/pmc_summary.txt
python3 tools/mfma_util.py $O/mfma $O/mfma_util.json > $O/mfma_util.txt || true
cp $(find $O/stats -name '*kernel_stats.csv' | head -1) $O/kernel_stats.csv
python3 tools/valu_util.py $O/valu > $O/valu_util.txt 2>&1 || true
find $O -name '*kernel_trace.csv' -delete; find $O -name '*agent_info.csv' -delete; find $O -name '*counter_collection.csv' -delete
python3 bench.py --route $RT > $O/bench_line.json 2> $O/bench.err
head -8 $O/kernel_stats.csv | cut -c1-170; cat $O/pmc_summary.txt; cat $O/mfma_util.txt; cat $O/valu_util.txt; tail -c 600 $O/bench_line.json
