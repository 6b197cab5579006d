"""BASELINE config C5 (n = 8192, k = 11, Rogers screened Coulomb, one channel) end to end; with --old the same solve
through the previous generation of kernels (sb2st v3, panel QR v1, pivot-recurrence bisection, one sy2sb group), so
that two independent implementations of every stage can be compared at a size no fixture covers."""
import os, sys, time, numpy as np
if "--old" in sys.argv:
    os.environ.update(BSP_SB2ST_VERSION="3", BSP_PANEL_QR="1", BSP_BISECT="1", BSP_SY2SB_GROUPS="1")
sys.path.insert(0, "/root/repo")
from bspatom_amd import capi
inp = capi.make_input(kind_grid=0, ra=0.0, rb=800.0, k=11, nfun=8192, n0_ini=1, l_ini=0, l_fin=0, zatom=20.0, kind_pot=1)
prob = capi.Problem(inp)
t0 = time.time(); E, info = prob.solve(0, 1); t1 = time.time()
E, info = prob.solve(0, 1); t2 = time.time()
print("n", prob.nfun, "info", info, "sorted", bool(np.all(np.diff(E[0]) >= 0)), "E[0..2]", E[0, :3], "E[-1]", E[0, -1],
      "solve %.2f s (first %.2f s)" % (t2 - t1, t1 - t0), "stages ms", prob.last_timing())
np.save("/root/repo/gpurun_out/c5_%s.npy" % ("old" if "--old" in sys.argv else "new"), E[0])
prob.close()
