#!/bin/bash
# round 4, first contact of the band route with the GPU: stage test, one full-size case both routes, a bench line each
set -o pipefail
mkdir -p gpurun_out/r4a
timeout -k 10 600 python -m pytest tests/test_gpu_stages.py -x -q -k "crawford" > gpurun_out/r4a/pytest_crawford.log 2>&1 || { tail -30 gpurun_out/r4a/pytest_crawford.log; exit 1; }
tail -3 gpurun_out/r4a/pytest_crawford.log
timeout -k 10 900 python -m pytest tests/test_gpu_solve.py -x -q -k "c3_1024 or c2_2048 or rydberg or spectra_vs_reference" > gpurun_out/r4a/pytest_solve.log 2>&1 || { tail -40 gpurun_out/r4a/pytest_solve.log; exit 1; }
tail -3 gpurun_out/r4a/pytest_solve.log
BSP_ROUTE=2 timeout -k 10 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r4a/bench_band.json 2> gpurun_out/r4a/bench_band.err || { tail -20 gpurun_out/r4a/bench_band.err; exit 1; }
cat gpurun_out/r4a/bench_band.json
