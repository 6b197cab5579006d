# LDS / issue counters of the kernels of one bench step (separate passes, --kernel-trace only) -> gpurun_out/lds/lds_util.json
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/lds; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $O/p1 -o a -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-timing --no-dense-leg --no-full-v > $O/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_WAVE_CYCLES --output-format csv -d $O/p2 -o b -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-timing --no-dense-leg --no-full-v > $O/p2.log 2>&1
cd $R
python3 tools/lds_util.py $O/p1 $O/p2 $O/lds_util.json > $O/lds_util.txt 2>&1
find $O -name '*kernel_trace.csv' -delete; find $O -name '*agent_info.csv' -delete; find $O -name '*counter_collection.csv' -delete
cat $O/lds_util.txt | head -14; tail -2 $O/p1.log | cut -c1-200
