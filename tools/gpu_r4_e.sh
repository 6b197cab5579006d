#!/bin/bash
# ratchet measurement of both routes against the COMMITTED file (only tightens; exits 1 if a figure is over its gate)
set -o pipefail
O=gpurun_out/r4e; mkdir -p $O
timeout -k 10 500 python tools/make_ratchet.py --route 1 > $O/ratchet_dense.log 2>&1; rc=$?
[ $rc -ge 124 ] && { tail -5 $O/ratchet_dense.log; exit $rc; }
timeout -k 10 500 python tools/make_ratchet.py --base gpurun_out/accuracy_ratchet.json "$@" > $O/ratchet_default.log 2>&1; rc=$?
[ $rc -ge 124 ] && { tail -5 $O/ratchet_default.log; exit $rc; }
grep -h "OVER THE GATE\|make_ratchet:" $O/ratchet_dense.log $O/ratchet_default.log | cut -c1-1100
cp gpurun_out/accuracy_ratchet.json $O/accuracy_ratchet_final.json
