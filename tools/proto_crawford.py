#!/usr/bin/env python3
"""proto_crawford.py -- dense NumPy statement of the band-preserving reduction of a banded pencil to standard form.

H c = E S c with H, S symmetric banded (half-width b = k - 1, matrices.f90:244-248), S = L L^T (banded Cholesky).
The dense route forms C = L^-1 H L^-T (dense, n^2 doubles) and tridiagonalises it in O(n^3).  L is a product of
block-row elementary factors, L = R_1 R_2 ... R_N (R_j = identity except block row j = [L_{j,j-1}, L_jj]); applying
R_j^-1 from both sides touches block row / column j only and leaves ONE b x b block of fill at (j, j-2), which is
chased off the top by orthogonal transformations that mix block columns (p, p+1) with p + 1 < j -- those commute with
every later R_k (Crawford 1973; blocked as in Lang 2019).  The result is block tridiagonal with full b x b blocks:
half-width 2b - 1 (15 for k = 9), O(n^2 b) flops, no dense matrix.

Index conventions fixed here (blocks 0-based, block size b, the last block ragged):
  step j      : row_j <- L_jj^-1 (row_j - L_{j,j-1} row_{j-1}), the same on columns; fill F at (j, j-2)
  item (j, s) : p = j - 2 - s;  RQ of [A(p+2, p), A(p+2, p+1)] = [0 R] Q^T;  columns and rows (p, p+1) <- Q;
                new fill at (p+1, p-1)
  items (j, s) and (j', s') with 2j + s = 2j' + s' touch disjoint blocks (checked below with a write log).

usage: python tools/proto_crawford.py [case] [channel ...]   (needs the oracle; compares with the truth fixtures)
"""
import os
import sys
import numpy as np
import scipy.linalg as sl

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
sys.path.insert(0, os.path.join(HERE, "..", "tests"))


EXPLICIT_INVERSE = os.environ.get("EXPLICIT_INVERSE", "1") == "1"


def dense_from_upper_band(B):
    k, n = B.shape
    M = np.zeros((n, n))
    for d in range(k):
        i = np.arange(n - d)
        M[i, i + d] = B[d, :n - d]
        M[i + d, i] = B[d, :n - d]
    return M


def crawford_block(S, H, b, log=None):
    """Returns the block tridiagonal standard-form matrix (dense storage) orthogonally similar to L^-1 H L^-T."""
    n = S.shape[0]
    L = np.linalg.cholesky(S)
    A = H.copy()
    N = (n + b - 1) // b
    blk = lambda j: slice(j * b, min((j + 1) * b, n))
    for j in range(N):
        rj = blk(j)
        if EXPLICIT_INVERSE:
            # what the kernel does: Li = L_jj^-1 and K = Li L_{j,j-1} formed once (S is the same for every channel), the step
            # itself is the congruence with Rinv = [I 0; -K Li] on block rows / columns (j-1, j)
            Li = sl.solve_triangular(L[rj, rj], np.eye(rj.stop - rj.start), lower=True)
            if j > 0:
                rm = blk(j - 1)
                K = Li @ L[rj, rm]
                A[rj, :] = Li @ A[rj, :] - K @ A[rm, :]
                A[:, rj] = A[:, rj] @ Li.T - A[:, rm] @ K.T
            else:
                A[rj, :] = Li @ A[rj, :]
                A[:, rj] = A[:, rj] @ Li.T
        else:
            if j > 0:
                rm = blk(j - 1)
                M = L[rj, rm]
                A[rj, :] -= M @ A[rm, :]
                A[:, rj] -= A[:, rm] @ M.T
            A[rj, :] = sl.solve_triangular(L[rj, rj], A[rj, :], lower=True)
            A[:, rj] = sl.solve_triangular(L[rj, rj], A[:, rj].T, lower=True).T
        if log is not None:
            log.append((2 * j - 1, j, -1, {("E", j - 2), ("E", j - 1), ("E", j), ("D", j), ("D", j - 1)}))
        for p in range(j - 2, -1, -1):
            c0, c1, r2 = blk(p), blk(p + 1), blk(p + 2)
            cols = slice(c0.start, c1.stop)
            X = A[r2, cols]
            Rf, Qs = sl.rq(X, mode="full")                 # X = Rf Qs, Rf = [0 R]
            Q = Qs.T
            A[:, cols] = A[:, cols] @ Q
            A[cols, :] = Q.T @ A[cols, :]
            A[r2, c0] = 0.0
            A[c0, r2] = 0.0
            if log is not None:
                log.append((2 * j + (j - 2 - p), j, j - 2 - p,
                            {("E", p + 1), ("D", p), ("E", p), ("D", p + 1), ("E", p - 1)}))
    return A


def check_wavefronts(log):
    """items of one wavefront t = 2 j + s (the elimination of step j at 2 j - 1) write disjoint blocks"""
    by_t = {}
    for t, j, s, blocks in log:
        by_t.setdefault(t, []).append((j, s, blocks))
    for t, items in by_t.items():
        seen = {}
        for j, s, blocks in items:
            for bl in blocks:
                assert bl not in seen, ("wavefront", t, "items", seen[bl], (j, s), "share", bl)
                seen[bl] = (j, s)
    return len(by_t), max(len(v) for v in by_t.values())


def upper_band(A, hb):
    n = A.shape[0]
    B = np.zeros((hb + 1, n))
    for d in range(hb + 1):
        i = np.arange(n - d)
        B[d, :n - d] = A[i, i + d]
    return B


def main():
    import oracle as orc
    from oracle import truth as qt
    from tests_truth import case_cfg
    case = sys.argv[1] if len(sys.argv) > 1 else "c3_1024_l31"
    chans = [int(x) for x in sys.argv[2:]] or [0, 14, 31]
    c = case_cfg(case)
    g = np.load(os.path.join(HERE, "..", "tests", "golden", case + ".npz"))
    tr = np.load(os.path.join(HERE, "..", "tests", "golden", "truth_" + case + ".npz"))
    rt, aind, xg, wg = orc.grid(c)
    nch = g["E"].shape[0]
    SB, HB = orc.assemble_bands(c, rt, aind, xg, wg, 0, nch)
    b = c.k - 1
    S = dense_from_upper_band(SB)
    # FLIP (default): the reduction runs on the index-reversed pencil, i.e. the fill is chased towards LARGE r, where the
    # entries of H are small; chasing it towards r = 0 (the centrifugal term, entries ~ l (l + 1) / r^2) costs the eigenvalues next
    # to zero a factor of 50 at l = 14 (0.015 against 0.0003 eps lambda_max)
    FLIP = os.environ.get("FLIP", "1") == "1"
    if FLIP:
        S = S[::-1, ::-1].copy()
    eps = np.finfo(float).eps
    for l in chans:
        H = dense_from_upper_band(HB[l])
        if FLIP:
            H = H[::-1, ::-1].copy()
        log = [] if l == chans[0] else None
        A = crawford_block(S, H, b, log)
        if log is not None:
            print("wavefronts %d, widest %d items" % check_wavefronts(log))
        n = A.shape[0]
        i, jx = np.indices(A.shape)
        out = np.max(np.abs(A[np.abs(i - jx) > 2 * b - 1]))
        asym = np.max(np.abs(A - A.T))
        sel = tr["chan"] == l
        idx = tr["idx"][sel]; truth = tr["hi"][sel]; ref = tr["ref"][sel]
        lam = float(np.max(np.abs(g["E"][l])))
        AB = upper_band((A + A.T) / 2, 2 * b - 1)
        IB = np.zeros_like(AB); IB[0] = 1.0
        hi, lo = qt.band_eigs(IB, AB, idx, truth, lam)
        err = np.abs(hi - truth)
        near = np.argsort(np.abs(truth))[:24]
        eref = np.abs(ref - truth)
        print("%s l=%d n=%d b=%d: outside band %.1e, asym %.1e | reduction error vs truth: worst rel %.2e, beyond 1e-10: %d, "
              "near zero %.4f eps lam  (reference LAPACK: %.2e, %d, %.4f)"
              % (case, l, n, b, out, asym, np.max(err / np.abs(truth)), np.sum(err > 1e-10 * np.abs(truth)),
                 np.max(err[near]) / (eps * lam), np.max(eref / np.abs(truth)), np.sum(eref > 1e-10 * np.abs(truth)),
                 np.max(eref[near]) / (eps * lam)))


if __name__ == "__main__":
    main()
