#!/usr/bin/env python3
"""proto_sbr.py -- dense NumPy prototype of the TWO-STEP band reduction planned for round 3:

    band 64  --(sb2sb: block bulge chasing, 16 columns per sweep)-->  band 16  --(single-column chasing)-->  tridiagonal

It exists to fix the index conventions the HIP kernels follow (csrc/sb2sb.hip) and to check them: which tiles an item reads and
writes, that the working band never exceeds 2b-1 sub-diagonals (the AB layout of csrc/sy2sb.hip::extract_band_kernel holds
exactly that), which items of consecutive sweeps may run at the same time (wavefront t = k + LAG * sweep), and that the
eigenvalues survive.  Nothing here is product code.

Item (sweep s, step k), j0 = NBLK*s, R_k = [j0 + D + B*k, j0 + D + B*(k+1)):
    k = 0: QR of A[R_0, j0:j0+NBLK]            (B-D+NBLK = B rows for D = NBLK)
    k > 0: QR of the first NBLK columns of the bulge block A[R_k, R_{k-1}], Q^T applied to its other B-NBLK columns
    then   D-tile:  A[R_k, R_k]     <- Q^T A[R_k, R_k] Q
           B'-tile: A[R_{k+1}, R_k] <- A[R_{k+1}, R_k] Q        (becomes the bulge block of step k+1)
"""
import sys
import numpy as np

B, D, NBLK, LAG = 64, 16, 16, 3


def house_qr(M):
    """Householder QR of M (m x c) -> V (unit lower trapezoidal, m x c), tau (c), R in the upper triangle of the result."""
    M = M.copy()
    m, c = M.shape
    V = np.zeros((m, c)); tau = np.zeros(c)
    for i in range(min(c, m)):
        x = M[i:, i]
        nrm = np.linalg.norm(x[1:])
        if nrm == 0.0:
            V[i, i] = 1.0
            continue
        beta = -np.copysign(np.hypot(x[0], nrm), x[0])
        tau[i] = (beta - x[0]) / beta
        v = x / (x[0] - beta); v[0] = 1.0
        V[i:, i] = v
        M[i:, i:] -= tau[i] * np.outer(v, v @ M[i:, i:])
    return V, tau, M


def form_T(V, tau):
    """Upper triangular T with Q = H_1 ... H_c = I - V T V^T (forward, columnwise: LAPACK dlarft)."""
    c = V.shape[1]
    T = np.zeros((c, c))
    for i in range(c):
        T[i, i] = tau[i]
        if i:
            T[:i, i] = -tau[i] * (T[:i, :i] @ (V[:, :i].T @ V[:, i]))
    return T


def item(A, s, k, stats):
    n = A.shape[0]
    j0 = NBLK * s
    r0 = j0 + D + B * k
    if r0 >= n:
        return False
    r1 = min(r0 + B, n)
    R = slice(r0, r1)
    if k == 0:
        cols = slice(j0, j0 + NBLK)
    else:
        cols = slice(r0 - B, r0 - B + NBLK)
    V, tau, Rm = house_qr(A[R, cols])
    T = form_T(V, tau)
    Q = np.eye(r1 - r0) - V @ T @ V.T
    # left side: rows R of the bulge block (k > 0: columns R_{k-1}; k = 0: the panel itself), mirrored into the upper triangle
    lc = cols if k == 0 else slice(r0 - B, r0)
    A[R, lc] = Q.T @ A[R, lc]
    A[R, cols] = np.triu(Rm[:, :]) if True else A[R, cols]
    A[lc, R] = A[R, lc].T
    # diagonal tile
    A[R, R] = Q.T @ A[R, R] @ Q
    A[R, R] = 0.5 * (A[R, R] + A[R, R].T)
    # next bulge block
    r2 = min(r1 + B, n)
    if r2 > r1:
        Rn = slice(r1, r2)
        A[Rn, R] = A[Rn, R] @ Q
        A[R, Rn] = A[Rn, R].T
    stats["items"] += 1
    return True


def bandwidth(A, tol=0.0):
    n = A.shape[0]
    bw = 0
    for d in range(n - 1, 0, -1):
        if np.max(np.abs(np.diag(A, -d))) > tol:
            bw = d
            break
    return bw


def sb2sb(A, check_order=True):
    """Block bulge chasing in WAVEFRONT order: all items (s, k) with k + LAG*s = t run 'at the same time' (here: on a snapshot
    check that their tile sets are disjoint), t ascending."""
    n = A.shape[0]
    S = (n - 2 * NBLK) // NBLK + 1                       # sweeps j0 = 0, 16, ..., n-32
    K = lambda s: max(0, -(-(n - (NBLK * s + D)) // B))  # items of sweep s
    stats = {"items": 0, "wavefronts": 0, "max_parallel": 0, "max_bw": 0}
    tmax = max(K(s) - 1 + LAG * s for s in range(S))
    for t in range(tmax + 1):
        active = [(s, t - LAG * s) for s in range(S) if 0 <= t - LAG * s < K(s)]
        if not active:
            continue
        if check_order:                                   # tile sets of concurrent items must not overlap: an item owns the
            boxes = []                                    # lower-triangle region rows [r0, r2) x columns [c0, r1)
            for s, k in active:
                r0 = NBLK * s + D + B * k
                boxes.append((r0, min(n, r0 + 2 * B), NBLK * s if k == 0 else r0 - B, min(n, r0 + B)))
            for i in range(len(boxes)):
                for j in range(i + 1, len(boxes)):
                    a, b = boxes[i], boxes[j]
                    assert not (a[0] < b[1] and b[0] < a[1] and a[2] < b[3] and b[2] < a[3]), ("overlap", t, a, b)
        for s, k in active:
            item(A, s, k, stats)
        stats["wavefronts"] += 1
        stats["max_parallel"] = max(stats["max_parallel"], len(active))
        stats["max_bw"] = max(stats["max_bw"], bandwidth(A, 0.0))
    return stats


def chase_to_tridiagonal(A, b):
    """Textbook single-column bulge chasing of a band-b matrix (dense storage) -- stage 2, b = D."""
    n = A.shape[0]
    for s in range(n - 2):
        r0 = s + 1
        c = s
        while r0 < n:
            r1 = min(r0 + b, n)
            x = A[r0:r1, c].copy()
            nrm = np.linalg.norm(x[1:])
            if nrm != 0.0:
                beta = -np.copysign(np.hypot(x[0], nrm), x[0])
                tau = (beta - x[0]) / beta
                v = x / (x[0] - beta); v[0] = 1.0
                H = np.eye(r1 - r0) - tau * np.outer(v, v)
                A[r0:r1, :] = H @ A[r0:r1, :]
                A[:, r0:r1] = A[:, r0:r1] @ H
            c = r0
            r0 = r0 + b
    return A


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 448
    rng = np.random.default_rng(1)
    A = np.zeros((n, n))
    for d in range(B + 1):
        v = rng.standard_normal(n - d) * (10.0 ** rng.uniform(-3, 0))
        A += np.diag(v, -d) + (np.diag(v, d) if d else 0)
    ev0 = np.linalg.eigvalsh(A)
    st = sb2sb(A)
    bw = bandwidth(A, 1e-13 * np.max(np.abs(ev0)))
    ev1 = np.linalg.eigvalsh(A)
    print("n=%d  sb2sb: %d items in %d wavefronts (max %d concurrent), working band <= %d sub-diagonals, result band %d,"
          " eigenvalue drift %.2e of lambda_max" % (n, st["items"], st["wavefronts"], st["max_parallel"], st["max_bw"], bw,
                                                      np.max(np.abs(ev1 - ev0)) / np.max(np.abs(ev0))))
    assert bw == D and st["max_bw"] <= 2 * B - 1
    A16 = np.triu(np.tril(A, D), -D)
    chase_to_tridiagonal(A16, D)
    bw2 = bandwidth(A16, 1e-13 * np.max(np.abs(ev0)))
    ev2 = np.linalg.eigvalsh(np.triu(np.tril(A16, 1), -1))
    print("stage 2: band %d, eigenvalue drift %.2e of lambda_max" % (bw2, np.max(np.abs(ev2 - ev0)) / np.max(np.abs(ev0))))
    assert bw2 == 1


if __name__ == "__main__":
    main()
