# regenerate the raw material of profiles/: kernel stats, two PMC passes (HBM bytes), MFMA utilisation, a bench line, micro-benchmarks
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-kernel-timing > $O/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o f -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-timing > $O/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o w -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-timing > $O/write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/mfma -o m -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-timing > $O/mfma.log 2>&1
cd $R
python3 tools/pmc_summary.py $O/fetch $O/write $O/pmc_summary.json 128 4096 > $O/pmc_summary.txt
python3 tools/mfma_util.py $O/mfma $O/mfma_util.json > $O/mfma_util.txt
cp $(find $O/stats -name '*kernel_stats.csv' | head -1) $O/kernel_stats.csv
find $O -name '*kernel_trace.csv' -delete; find $O -name '*agent_info.csv' -delete; find $O -name '*counter_collection.csv' -delete
(make -C tools/microbench > /dev/null 2>&1 && cd tools/microbench && for w in 1 2 4; do echo "# mfma_f64_peak, $w workgroup(s) of 4 waves per CU"; ./mfma_f64_peak $w; done; echo "# gemm_core"; ./gemm_core; echo "# tile_bw"; ./tile_bw; echo "# sturm_rate (512 threads)"; ./sturm_rate 512) > $O/microbench.txt 2>&1
python3 bench.py > $O/bench_line.json 2> $O/bench.err
head -12 $O/kernel_stats.csv | cut -c1-160; cat $O/pmc_summary.txt; cat $O/mfma_util.txt; tail -c 900 $O/bench_line.json
