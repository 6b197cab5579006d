#!/usr/bin/env python3
"""dump_tridiagonal.py -- the tridiagonal matrices (d, e) the band route hands to the bisection, for a few channels of a golden case, into
gpurun_out/tri_<case>.npz (GPU): material for models of the bisection's starting grid (tools/sim_grid.py).
usage: python tools/dump_tridiagonal.py [case [channel ...]]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bspatom_amd import capi
from bspatom_amd.namelist import read_namelists

case = sys.argv[1] if len(sys.argv) > 1 else "c4_4096"
chans = [int(a) for a in sys.argv[2:]] or [0, 1, 30, 64, 127]
nl = read_namelists(open(os.path.join(ROOT, "tests", "golden", "inputs", case + ".inp")).read())
kw = {}
kw.update(nl["vars_bsp"]); kw.update(nl["vars_tise"])
prob = capi.Problem(capi.make_input(**kw))
chans = [c for c in chans if c <= prob.lmax]
out = {}
for l in chans:
    SB, HB = prob.assemble(l, 1)
    AB, info = capi.stage_crawford(SB, HB)
    assert info == 0
    d, e = capi.stage_sb2st(AB, prob.nfun)
    out["d%d" % l] = d[0]; out["e%d" % l] = e[0]
    print(case, "l", l, "n", prob.nfun, "d range", d.min(), d.max(), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.savez(os.path.join(ROOT, "gpurun_out", "tri_%s.npz" % case), chans=np.array(chans), **out)
