"""Runs the BASELINE.json configs C3 (l=0..31, n=2048) and C5 (Rogers/Yukawa-type, n=8192, k=11) on the GPU and
checks them against size-independent properties (Rydberg series, S-orthonormality / residual of an eigenvector)."""
import sys, time, numpy as np
sys.path.insert(0, ".")
from bspatom_amd import capi
t0 = time.time()
p = capi.Problem(capi.make_input(kind_grid=0, rb=400.0, k=9, nfun=2048, l_fin=31))
E, info = p.solve(0, 32)
print("C3 n=2048 l=0..31: info ok", (info == 0).all(), "timing", {k: round(v, 1) for k, v in p.last_timing().items()})
for l in (0, 5, 31):
    n0 = l + 1
    ex = -0.5 / np.arange(n0, n0 + 4) ** 2
    print("  l=%d E[:4]=%s  rydberg rel err %.1e" % (l, E[l, :4], np.max(np.abs(E[l, :4] - ex) / np.abs(ex))))
p.close()
p = capi.Problem(capi.make_input(kind_grid=0, rb=800.0, k=11, nfun=8192, l_fin=0, zatom=20.0, kind_pot=1))
E, info = p.solve(0, 1)
print("C5 n=8192 k=11 KIND_POT=1: info", info, "E[:3]", E[0, :3], "Emax", E[0, -1], "timing", {k: round(v, 1) for k, v in p.last_timing().items()})
c = p.eigvec(0, 1)
SB, HB = p.assemble(0, 1)
n, k = p.nfun, p.k
def bmv(Bd, x):
    y = Bd[0] * x
    for d in range(1, k):
        y[:n - d] += Bd[d, :n - d] * x[d:]; y[d:] += Bd[d, :n - d] * x[:n - d]
    return y
print("  eigvec: c^T S c - 1 = %.1e, residual/|E| = %.1e" % (c @ bmv(SB, c) - 1, np.max(np.abs(bmv(HB[0], c) - E[0, 0] * bmv(SB, c))) / abs(E[0, 0])))
print("  sorted:", bool(np.all(np.diff(E[0]) >= 0)), " wall %.1fs" % (time.time() - t0))
