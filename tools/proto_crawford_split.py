#!/usr/bin/env python3
"""proto_crawford_split.py -- the band-preserving reduction of tools/proto_crawford.py run from BOTH ends of the pencil.

The one-sided reduction chases the fill of elimination step j over j blocks: N^2 / 2 chase items.  Cut the pencil in the middle,
n = n1 + n2:  the leading part (H11, S11) and the index-reversed trailing part (H22, S22) are reduced independently, each by the
one-sided process with its block 0 at its own END of the matrix (fill is chased away from the middle): 2 (N/2)^2 / 2 = N^2 / 4 items.
What is left of S is [I G; G^T I] with ONE b x b block G = L1_last^-1 S(m-1, m) L2_last^-T next to the cut (the chase never touches
the two blocks next to the cut), which one more elimination step at the cut removes; its fill is chased to one end: N / 2 items.

usage: python tools/proto_crawford_split.py [case] [channel ...]    (needs the oracle; accuracy against the truth fixtures)
"""
import os
import sys
import numpy as np
import scipy.linalg as sl

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
sys.path.insert(0, os.path.join(HERE, "..", "tests"))
sys.path.insert(0, HERE)
import proto_crawford as pc


def one_sided(A, Sm, L, b, nsub, j_from=0, count=None):
    """Process A on the leading nsub x nsub part of the full matrices A (and Sm, carried along as a check): elimination steps
    j_from .. with the factor blocks of L (nsub x nsub, lower), the fill chased towards index 0.  Rows and columns beyond nsub take
    part in the row / column operations like any other."""
    N = (nsub + b - 1) // b
    blk = lambda j: slice(j * b, min((j + 1) * b, nsub))
    items = 0
    for j in range(j_from, N):
        rj = blk(j)
        Li = sl.solve_triangular(L[rj, rj], np.eye(rj.stop - rj.start), lower=True)
        for M in (A, Sm):
            if j > 0:
                rm = blk(j - 1)
                K = Li @ L[rj, rm]
                M[rj, :] = Li @ M[rj, :] - K @ M[rm, :]
                M[:, rj] = M[:, rj] @ Li.T - M[:, rm] @ K.T
            else:
                M[rj, :] = Li @ M[rj, :]
                M[:, rj] = M[:, rj] @ Li.T
        for p in range(j - 2, -1, -1):
            c0, c1, r2 = blk(p), blk(p + 1), blk(p + 2)
            cols = slice(c0.start, c1.stop)
            Rf, Qs = sl.rq(A[r2, cols], mode="full")
            Q = Qs.T
            for M in (A, Sm):
                M[:, cols] = M[:, cols] @ Q
                M[cols, :] = Q.T @ M[cols, :]
            A[r2, c0] = 0.0
            A[c0, r2] = 0.0
            items += 1
    if count is not None:
        count[0] += items


def crawford_split(S, H, b, n1, count=None):
    n = S.shape[0]
    n2 = n - n1
    assert n1 % b == 0 and n2 % b == 0, "both parts whole blocks (the cut and both ends on block boundaries)"
    A = H.copy(); Sm = S.copy()
    # leading part: blocks from index 0, chase towards index 0
    one_sided(A, Sm, np.linalg.cholesky(S[:n1, :n1]), b, n1, count=count)
    # trailing part: the same on the index-reversed matrices (its block 0 is the last block of the matrix)
    Ar = A[::-1, ::-1].copy(); Sr = Sm[::-1, ::-1].copy()
    one_sided(Ar, Sr, np.linalg.cholesky(S[n1:, n1:][::-1, ::-1]), b, n2, count=count)
    # what is left of S: identity + one block next to the cut.  In the reversed numbering the cut lies behind block n2 / b - 1
    G = Sr - np.eye(n)
    mask = np.ones((n, n), bool)
    mask[n2 - b:n2, n2:n2 + b] = False; mask[n2:n2 + b, n2 - b:n2] = False
    assert np.max(np.abs(G[mask])) < 1e-9 * np.max(np.abs(S)), np.max(np.abs(G[mask]))
    # the elimination step at the cut, in the reversed numbering (block j = n2 / b takes the factor of [I G; G^T I]), its fill chased
    # towards the reversed index 0 = the END of the matrix
    Lc = np.linalg.cholesky(Sr[:n2 + b, :n2 + b])
    one_sided(Ar, Sr, Lc, b, n2 + b, j_from=n2 // b, count=count)
    assert np.max(np.abs(Sr - np.eye(n))) < 1e-9 * np.max(np.abs(S))
    return Ar[::-1, ::-1]


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
    S = pc.dense_from_upper_band(SB)
    n = S.shape[0]
    frac = float(os.environ.get("CUT", "0.5"))
    n1 = int(round(n * frac / b)) * b
    eps = np.finfo(float).eps
    for l in chans:
        H = pc.dense_from_upper_band(HB[l])
        cnt = [0]
        A = crawford_split(S, H, b, n1, cnt)
        i, jx = np.indices(A.shape)
        out = np.max(np.abs(A[np.abs(i - jx) > 2 * b - 1]))
        # half-width b everywhere but the two end blocks
        inner = (np.abs(i - jx) > b) & (np.minimum(i, jx) >= b) & (np.maximum(i, jx) < n - b)
        sel = tr["chan"] == l
        idx = tr["idx"][sel]; truth = tr["hi"][sel]; ref = tr["ref"][sel]
        lam = float(np.max(np.abs(g["E"][l])))
        AB = pc.upper_band((A + A.T) / 2, 2 * b - 1)
        IB = np.zeros_like(AB); IB[0] = 1.0
        hi, lo = qt.band_eigs(IB, AB, idx, truth, lam)
        err = np.abs(hi - truth)
        near = np.argsort(np.abs(truth))[:24]
        eref = np.abs(ref - truth)
        print("%s l=%d n=%d cut at %d, %d chase items (one-sided: %d): outside 2b-1 %.1e, outside b away from the ends %.1e | worst rel %.2e, "
              "beyond 1e-10: %d, near zero %.4f eps lam  (reference LAPACK: %.2e, %d, %.4f)"
              % (case, l, n, n1, cnt[0], (n // b - 1) * (n // b - 2) // 2, out, np.max(np.abs(A[inner])), np.max(err / np.abs(truth)),
                 np.sum(err > 1e-10 * np.abs(truth)), np.max(err[near]) / (eps * lam), np.max(eref / np.abs(truth)),
                 np.sum(eref > 1e-10 * np.abs(truth)), np.max(eref[near]) / (eps * lam)))


if __name__ == "__main__":
    main()
