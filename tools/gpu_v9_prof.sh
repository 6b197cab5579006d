# kernel statistics of one bench step with the two-step band reduction (BSP_SB2ST_VERSION=9)
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof9; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export BSP_SB2ST_VERSION=9   # also the default for n >= 512
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/stats.log 2>&1
cp $(find $O/stats -name '*kernel_stats.csv' | head -1) $O/kernel_stats.csv
find $O -name '*kernel_trace.csv' -delete; find $O -name '*agent_info.csv' -delete
head -8 $O/kernel_stats.csv | cut -c1-200
