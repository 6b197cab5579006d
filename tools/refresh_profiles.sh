# regenerates the raw material of profiles/: kernel stats of `bench.py --steps 4 --warmup 1`, two PMC passes, a bench line
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline > $O/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o f -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o w -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/write.log 2>&1
cd $R && python3 bench.py > $O/bench_line.json 2> $O/bench.err
find $O -name '*kernel_trace.csv' -delete; find $O -name '*agent_info.csv' -delete
ls -la $O $O/*; tail -c 600 $O/bench_line.json
