#!/usr/bin/env python3
"""make_truth.py -- high-precision TRUTH eigenvalues for the parity tests (tests/golden/truth_*.npz).

TEST INFRASTRUCTURE.  For a namelist in tests/golden/inputs/ the oracle assembles the banded pencil (S, H_l) in
double precision -- bit-identical to what the compiled reference hands to DSYGV (matrices.f90:244-248; pinned in
tests/test_oracle_golden.py) -- and oracle/truth_quad.c finds selected eigenvalues of EXACTLY those matrices by
bisection on the inertia of H_l - x S (negative pivots of the banded LDL^T) in 113-bit arithmetic.  Nothing of the
algorithms under test (Cholesky, standard form, tridiagonalisation, Sturm bisection in doubles, LAPACK) is involved,
so the fixture measures the error of the reference's LAPACK spectrum and of the GPU spectrum separately:

    err_ref = |E_ref - truth|,  err_gpu = |E_gpu - truth|.

Stored per case: channel numbers, eigenvalue indices (0-based), truth as hi + lo doubles, and the reference's value
(from the case's golden fixture) for convenience.  Selected eigenvalues per channel: the NZ nearest zero (where a
relative bar is hardest; for the cases in GROW that set is grown until the reference's own error is small at its edge), the lowest NL, and NS spread over the rest of the spectrum.

--verify repeats a few of them with an independent 40-digit mpmath implementation of the same count.

usage: python tests/golden/make_truth.py [--verify] [case ...]
"""
import os
import subprocess
import sys
import time
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import oracle as orc                                   # noqa: E402
from oracle import truth as qt                         # noqa: E402
from bspatom_amd.namelist import read_namelists       # noqa: E402

NZ, NL, NS = 24, 8, 16

# case -> channels (None = all of the golden fixture's)
CASES = {
    "c3_1024_l31": None,
    "c4_4096": None,
    "c5_1024_k11": None,
    "c2_2048": None,
    "lin1024": None,
    "lin256": None,
    "c1_lin": None,
    "simfues": None, "bc1": None, "ka_ra": None, "yuk256": None,          # the other small linear-grid cases
    "bc1_2048": None, "sf2048": None, "exp2048": None, "explin2048": None,  # SURVEY 8(f).4 at scale
    "c5_8192": None,                                                       # BASELINE configs[4] at its real size
    "c3_2048_l31": None,                                                   # BASELINE configs[2] at its real size, all 32 channels
    "c4_4096_l127": None,                                                  # BASELINE configs[3]: all 128 channels of the bench workload
}


def case_cfg(name, text=None):
    if text is None:
        text = open(os.path.join(HERE, "inputs", name + ".inp")).read()
    nl = read_namelists(text)
    kw = {}
    kw.update(nl["vars_bsp"]); kw.update(nl["vars_tise"])
    return orc.make_cfg(**kw)


# cases whose near-zero set is GROWN until the reference's own LAPACK error has dropped below EDGE relative at the
# outermost 8 members on either side: with 128 channels at n=4096 that noise (1e-12 .. 1e-11 absolute, growing with l)
# exceeds 1e-10 relative for many more than the NZ nearest eigenvalues, and every such eigenvalue needs its truth for
# the parity bar to tell the reference's error from the GPU's (tests/test_gpu_solve.py::full_size_bar)
GROW = {"c4_4096_l127"}
EDGE, STEP, CAP = 3e-11, 16, 640


def select(Eref):
    n = len(Eref)
    near = np.argsort(np.abs(Eref))[:NZ]
    low = np.arange(min(NL, n))
    spread = np.linspace(0, n - 1, NS).astype(int)
    return np.unique(np.concatenate([near, low, spread])).astype(np.int32)


def mp_count(SB, HBl, x, digits=40):
    """Independent restatement of the inertia count with mpmath (dense-window LDL^T, same pivot order)."""
    import mpmath as mp
    mp.mp.dps = digits
    k, n = SB.shape
    p = k - 1
    x = mp.mpf(x) if not isinstance(x, mp.mpf) else x
    # rows of the factor: keep the last p columns' (l_ij * d_j) products
    A = lambda i, j: (mp.mpf(float(HBl[j - i, i])) - x * mp.mpf(float(SB[j - i, i]))) if 0 <= j - i < k else mp.mpf(0)
    dpiv = []
    Lcol = []                                            # Lcol[j][r] = L(j+1+r, j), r = 0..p-1
    cnt = 0
    for j in range(n):
        d = A(j, j)
        for i in range(max(0, j - p), j):
            lji = Lcol[i][j - i - 1]
            d -= lji * lji * dpiv[i]
        if d == 0:
            d = mp.mpf("-1e-300")
        if d < 0:
            cnt += 1
        col = []
        for r in range(j + 1, min(n, j + p + 1)):
            v = A(j, r)
            for i in range(max(0, r - p), j):
                v -= Lcol[i][r - i - 1] * Lcol[i][j - i - 1] * dpiv[i]
            col.append(v / d)
        while len(col) < p:
            col.append(mp.mpf(0))
        dpiv.append(d); Lcol.append(col)
    return cnt


def mp_eig(SB, HBl, m, est, lam):
    import mpmath as mp
    mp.mp.dps = 40
    w = mp.mpf(64 * np.finfo(float).eps * lam)
    lo, hi = mp.mpf(est) - w, mp.mpf(est) + w
    while not (mp_count(SB, HBl, lo) <= m and mp_count(SB, HBl, hi) >= m + 1):
        w *= 4; lo -= w; hi += w
    while hi - lo > mp.mpf("1e-22") * abs((lo + hi) / 2) + mp.mpf(1e-30 * lam):
        mid = (lo + hi) / 2
        if mp_count(SB, HBl, mid) > m:
            hi = mid
        else:
            lo = mid
    return (lo + hi) / 2


def run_case(name, verify=False):
    g = np.load(os.path.join(HERE, name + ".npz"), allow_pickle=False)
    Eref = g["E"]
    nch = Eref.shape[0]
    c = case_cfg(name)
    rt, aind, xg, wg = orc.grid(c)
    SB, HB = orc.assemble_bands(c, rt, aind, xg, wg, 0, nch)
    chans, idxs, his, los, refs = [], [], [], [], []
    t0 = time.time()
    for l in range(nch):
        lam = float(np.max(np.abs(Eref[l])))
        idx = select(Eref[l])
        hi, lo = qt.band_eigs(SB, HB[l], idx, Eref[l][idx], lam)
        if name in GROW:
            known = dict(zip(idx.tolist(), zip(hi.tolist(), lo.tolist())))
            near = np.sort(np.argsort(np.abs(Eref[l]))[:NZ])
            a, b = int(near[0]), int(near[-1])                       # contiguous index range around zero
            while True:
                rel = lambda js: max(abs(Eref[l][j] - known[j][0]) / abs(known[j][0]) for j in js)
                up = b < c.nfun - 1 and b - a < CAP and rel(range(max(a, b - 7), b + 1)) > EDGE
                dn = a > 0 and b - a < CAP and rel(range(a, min(b, a + 7) + 1)) > EDGE
                if not (up or dn):
                    break
                a2 = max(0, a - STEP) if dn else a
                b2 = min(c.nfun - 1, b + STEP) if up else b
                new = np.array([j for j in range(a2, b2 + 1) if j not in known], dtype=np.int32)
                h2, l2 = qt.band_eigs(SB, HB[l], new, Eref[l][new], lam)
                known.update(zip(new.tolist(), zip(h2.tolist(), l2.tolist())))
                a, b = a2, b2
            idx = np.array(sorted(known), dtype=np.int32)
            hi = np.array([known[j][0] for j in idx]); lo = np.array([known[j][1] for j in idx])
        chans.append(np.full(len(idx), l, dtype=np.int32)); idxs.append(idx); his.append(hi); los.append(lo)
        refs.append(Eref[l][idx])
        err = np.abs(Eref[l][idx] - hi)
        rel = err / np.abs(hi)
        print("%-12s l=%2d n=%d: %d eigenvalues, reference error vs truth: worst rel %.2e (E=%.3e), worst abs %.2e = %.4f eps*lam_max"
              "  [%.0f s]" % (name, l, c.nfun, len(idx), rel.max(), hi[np.argmax(rel)], err.max(),
                              err.max() / (np.finfo(float).eps * lam), time.time() - t0), flush=True)
        if verify and l in (0, nch - 1) and c.nfun <= 1024:
            j = int(np.argmin(np.abs(hi)))
            import mpmath as mp
            v = mp_eig(SB, HB[l], int(idx[j]), float(Eref[l][idx[j]]), lam)
            d = abs(v - (mp.mpf(float(hi[j])) + mp.mpf(float(lo[j]))))
            print("    mpmath 40 digits, eigenvalue %d: %s   |quad - mpmath| = %s" % (idx[j], mp.nstr(v, 25), mp.nstr(d, 3)), flush=True)
            assert d <= mp.mpf("1e-20") * abs(v) + mp.mpf(1e-28 * lam)
    np.savez_compressed(os.path.join(HERE, "truth_" + name + ".npz"),
                        chan=np.concatenate(chans), idx=np.concatenate(idxs), hi=np.concatenate(his),
                        lo=np.concatenate(los), ref=np.concatenate(refs), sizes=np.array([c.nfun, c.k, nch]),
                        method=np.array("float128 bisection on LDL^T inertia of the oracle's bit-exact bands; rtol 1e-24"))


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    verify = "--verify" in sys.argv
    for name in (args or list(CASES)):
        run_case(name, verify)
