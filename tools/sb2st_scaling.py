"""Stage times of the batched solve against the matrix size (128 channels): the algorithmic work of bulge chasing
goes as n^2, of sy2sb as n^3; the departure from that is what start-up, tails and short sweeps cost."""
import sys, numpy as np
sys.path.insert(0, "/root/repo")
from bspatom_amd import capi
nl = 128
for n in (512, 1024, 2048, 3072, 4096):
    prob = capi.Problem(capi.make_input(kind_grid=0, ra=0.0, rb=800.0 * n / 4096, k=9, nfun=n, l_fin=nl - 1, zatom=1.0))
    prob.solve(0, nl)
    prob.solve(0, nl)
    ms = prob.last_timing()
    print("n=%5d  stage ms: %s" % (n, {k: round(v, 2) for k, v in ms.items()}), flush=True)
    prob.close()
