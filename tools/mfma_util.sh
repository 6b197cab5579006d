# MFMA pipe utilisation per kernel (counter pass; rocprofv3 serialises the kernels, so these are standalone figures)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/mfma; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O -o m -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/log.txt 2>&1
find $O -name '*kernel_trace.csv' -delete; find $O -name '*agent_info.csv' -delete
ls -la $O; tail -2 $O/log.txt | cut -c1-300
