#!/usr/bin/env python3
"""sim_grid.py -- the lock-step rounds of csrc/tridiag.hip::bisect3_kernel in an exact-arithmetic model (as tools/sim_secant.py: the
spectrum stands in for the matrix), a WORKGROUP at a time, to compare what an eigenvalue does while its bracket still holds several:
  bisect : the midpoint (rounds 1 - 4 of the kernel as it was)
  share  : the q eigenvalues of a bracket place q points in it, rank by rank, and every one of them takes the tightest bracket that
           ALL q counts allow (the points of a round are in LDS anyway)
Prints, per workgroup of 1024 eigenvalues, the number of unfinished eigenvalues after every round.
usage: python tools/sim_grid.py [golden case [channel]]   (tests/golden/<case>.npz with E; or gpurun_out/tri_<case>.npz from dump_tridiagonal.py)"""
import numpy as np, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
eps = 2.220446049250313e-16
NG = 1024

def spectrum(case, l):
    p = os.path.join(ROOT, "gpurun_out", "tri_%s.npz" % case)
    if os.path.exists(p):
        from scipy.linalg import eigvalsh_tridiagonal
        g = np.load(p)
        return eigvalsh_tridiagonal(g["d%d" % l], g["e%d" % l])
    return np.sort(np.load(os.path.join(ROOT, "tests", "golden", case + ".npz"))["E"][l])

def final(lo, hi):
    mid = 0.5 * (lo + hi)
    return (mid <= lo) | (mid >= hi) | (hi - lo <= 2 * eps * np.maximum(np.abs(lo), np.abs(hi)) + 1e-300)

def simulate(lam, mode, wg):
    n = len(lam); nrm = max(abs(lam[0]), abs(lam[-1])); lam = lam / nrm
    gl = lam[0] - 1e-3; gu = lam[-1] + 1e-3; w = gu - gl
    count = lambda x: np.searchsorted(lam, x, side="left")
    def logf(x):
        return np.round(np.array([np.sum(np.log2(np.abs(lam - xi) + 1e-300)) for xi in x]) * 2048) / 2048
    grid = gl + w * (np.arange(NG) + 1) / (NG + 1)
    cg = count(grid); lfg = logf(grid)
    m0 = wg * NG; m = np.arange(m0, min(m0 + NG, n)); K = len(m)
    R = np.searchsorted(cg, m, side="right"); L = R - 1
    lo = np.where(L < 0, gl, grid[np.maximum(L, 0)]); hi = np.where(R >= NG, gu, grid[np.minimum(R, NG - 1)])
    clo = np.where(L < 0, 0, cg[np.maximum(L, 0)]); chi = np.where(R >= NG, n, cg[np.minimum(R, NG - 1)])
    flo = np.where(L < 0, np.inf, lfg[np.maximum(L, 0)]); fhi = np.where(R >= NG, np.inf, lfg[np.minimum(R, NG - 1)])
    wref = np.full(K, np.inf); since = np.zeros(K, int); last = np.zeros(K, int)
    evals = np.ones(K, int); left = []
    for it in range(200):
        done = final(lo, hi)
        left.append(int(np.sum(~done)))
        if left[-1] == 0: break
        wd = hi - lo
        halved = wd <= 0.5 * wref
        wref = np.where(halved, wd, wref); since = np.where(halved, 0, since)
        slow = since >= 3; since = since + 1
        x = 0.5 * (lo + hi); sec = np.zeros(K, bool)
        iso = (chi - clo == 1) & np.isfinite(flo) & np.isfinite(fhi) & ~slow & ~done
        dl = np.clip(np.where(iso, flo - fhi, 0.0), -60, 60); r = 2.0 ** dl
        tiny = 2 * eps * np.maximum(np.abs(lo), np.abs(hi))
        xe = np.minimum(np.maximum(lo + wd * (r / (1 + r)), lo + tiny), hi - tiny)
        ok = iso & (xe > lo) & (xe < hi)
        x = np.where(ok, xe, x); sec = ok
        a = np.maximum(clo, m0); b = np.minimum(chi, m0 + K); q = b - a
        shared = (mode == "share") & (q > 1) & ~done & ~slow
        xs = lo + wd * ((m - a + 1) / (q + 1.0))
        x = np.where(shared & (xs > lo) & (xs < hi), xs, x)
        c = count(x); f = logf(x); evals += ~done
        up = c > m
        upd = ~done
        # Illinois
        flo2 = np.where(upd & up & sec & (last == 1), flo - 1, flo); fhi2 = np.where(upd & ~up & sec & (last == 2), fhi - 1, fhi)
        nhi = np.where(upd & up, x, hi); nchi = np.where(upd & up, c, chi); nfhi = np.where(upd & up, f, fhi2)
        nlo = np.where(upd & ~up, x, lo); nclo = np.where(upd & ~up, c, clo); nflo = np.where(upd & ~up, f, flo2)
        last = np.where(upd, np.where(sec, np.where(up, 1, 2), 0), last)
        if mode == "share":
            for i in np.nonzero(shared)[0]:
                s0, s1 = a[i] - m0, b[i] - m0
                cs = c[s0:s1]
                j = np.searchsorted(cs, m[i], side="right")          # first slot with count > m
                if j < len(cs) and nlo[i] < x[s0 + j] < nhi[i]: nhi[i], nchi[i], nfhi[i] = x[s0 + j], cs[j], f[s0 + j]
                if j > 0 and nlo[i] < x[s0 + j - 1] < nhi[i]: nlo[i], nclo[i], nflo[i] = x[s0 + j - 1], cs[j - 1], f[s0 + j - 1]
        lo, hi, clo, chi, flo, fhi = nlo, nhi, nclo, nchi, nflo, nfhi
    return evals, left

if __name__ == "__main__":
    case = sys.argv[1] if len(sys.argv) > 1 else "c4_4096"
    l = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    lam = spectrum(case, l)
    for mode in ("bisect", "share"):
        for wg in range((len(lam) + NG - 1) // NG):
            ev, left = simulate(lam, mode, wg)
            r8 = next(i for i, k in enumerate(left) if k <= NG // 8)
            print("%s l %d %-6s wg %d: evaluations mean %.1f median %d max %d; rounds until 1/8 are left %d; unfinished after round r:" % (
                case, l, mode, wg, ev.mean(), np.median(ev), ev.max(), r8), left[:40], flush=True)
