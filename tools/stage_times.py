#!/usr/bin/env python3
"""stage_times.py -- stage times of the solve for BASELINE configs[3]'s pencil (n = 4096, k = 9) under chosen switches; results are
not checked (timing experiments may compute garbage).  usage: tools/stage_times.py [--channels N] [--reps R] [name=value ...]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bspatom_amd import capi

args = sys.argv[1:]
chans, reps, opts = 128, 3, {}
while args:
    a = args.pop(0)
    if a == "--channels": chans = int(args.pop(0))
    elif a == "--reps": reps = int(args.pop(0))
    else:
        k, v = a.split("="); opts[k] = int(v)
for k, v in opts.items():
    capi.set_option(k, v)
prob = capi.Problem(capi.make_input(kind_grid=0, ra=0.0, rb=800.0, k=9, nfun=4096, n0_ini=1, l_ini=0, l_fin=127, zatom=1.0))
prob.solve(0, chans)
acc = {}
for _ in range(reps):
    prob.solve(0, chans)
    for k, v in prob.last_timing().items():
        acc[k] = acc.get(k, 0.0) + v / reps
print("%3d channels %-40s route %d: %s" % (chans, " ".join("%s=%d" % kv for kv in opts.items()), prob.route(),
                                           "  ".join("%s %.1f" % (k, v) for k, v in acc.items())))
prob.close()
