#!/bin/bash
# round 4: ratchet measurement with the final kernels: dense route first (it seeds max_abs and carries the one override), then the default
set -o pipefail
O=gpurun_out/r4e; mkdir -p $O
timeout -k 10 500 python tools/make_ratchet.py --route 1 --allow-regress "sf2048:near_zero:dense route only (the band route, the default for this case, measures 0.025): l = 0, eigenvalue 6, E = -1.0797e-2 is 4.66e-14 off the truth (0.044 eps lambda_max) since round 3's sb16r_kernel, which sums the products of a chase item inside a lane instead of across 16 lanes; sb16st_kernel measured 0.0195; the reference's LAPACK is 2.16e-13 off next to zero in the same channel" > $O/ratchet_dense.log 2>&1; rc=$?
[ $rc -ge 124 ] && { tail -5 $O/ratchet_dense.log; exit $rc; }
cp gpurun_out/accuracy_ratchet.json $O/accuracy_ratchet_after_dense.json
timeout -k 10 500 python tools/make_ratchet.py --base gpurun_out/accuracy_ratchet.json "$@" > $O/ratchet_default.log 2>&1; rc=$?
[ $rc -ge 124 ] && { tail -5 $O/ratchet_default.log; exit $rc; }
grep -h "OVER THE GATE\|make_ratchet:" $O/ratchet_dense.log $O/ratchet_default.log | cut -c1-900
cp gpurun_out/accuracy_ratchet.json $O/accuracy_ratchet_final.json
