"""Dense NumPy prototype of the panel factorisation of csrc/tsqr.hip (round 3): communication-avoiding QR of an m x 64 panel
(TSQR: row blocks of 256, radix-4 tree of stacked R factors, every node keeps its EXPLICIT 256 x 64 Q) followed by the
Householder reconstruction of Ballard, Demmel, Grigori, Jacquelin, Nguyen, Solomonik (IPDPS 2014): modified LU of Q1 - S on
the top 64 x 64 block gives V (unit lower trapezoidal), T (upper triangular) with  I - V T V^T  orthogonal and
(I - V T V^T)^T P = [R; 0].  Fixes the index conventions before any kernel is written:
  * node at level l, index i stacks the R factors of children 4 i .. 4 i + 3 of level l - 1 (missing children: zero blocks);
  * rows of Q1 owned by leaf i:  Q_leaf(i) . B(i),  B(i) = Qnode[l1](rows of child slot k0) . Qnode[l2](rows of slot k1) ... Qroot(rows ..)
  * S' (signs) is chosen during the LU: s_j = -sgn(x_jj) on the partially eliminated diagonal, pivot x_jj - s_j (|pivot| >= 1);
  * V = (Q1 - E S') U^-1,  T = -U S' L^-T  (L = top block of V),  R_final = S' R_tsqr,  W = V T.
Run: python tools/proto_tsqr.py"""
import numpy as np

NB, MB, RAD = 64, 256, 4


def house_qr_inplace(a):
    """Unblocked Householder QR as the kernel does it: returns R (NB x NB), explicit Q (rows x NB) by in-place back-accumulation
    (LAPACK dorg2r), with H = I when nothing is below the pivot."""
    a = a.copy()
    m = a.shape[0]
    taus = np.zeros(NB)
    for j in range(NB):
        alpha = a[j, j]
        sigma = float(np.dot(a[j + 1:, j], a[j + 1:, j]))
        if sigma == 0.0 or not (alpha * alpha + sigma > 1e-280):
            taus[j] = 0.0
            a[j + 1:, j] = 0.0
            continue
        beta = -np.copysign(np.sqrt(alpha * alpha + sigma), alpha)
        taus[j] = (beta - alpha) / beta
        a[j + 1:, j] *= 1.0 / (alpha - beta)
        a[j, j] = beta
        v = np.concatenate([[1.0], a[j + 1:, j]])
        w = taus[j] * (v @ a[j:, j + 1:])
        a[j:, j + 1:] -= np.outer(v, w)
    R = np.triu(a[:NB, :])
    # explicit Q in place, last reflector first
    for j in range(NB - 1, -1, -1):
        v = np.concatenate([[1.0], a[j + 1:, j]])
        if j + 1 < NB:
            w = taus[j] * (v @ a[j:, j + 1:])
            a[j:, j + 1:] -= np.outer(v, w)
        a[j + 1:, j] = -taus[j] * v[1:]
        a[j, j] = 1.0 - taus[j]
        a[:j, j] = 0.0
    return R, a


def tsqr_hr(P):
    m = P.shape[0]
    L = -(-m // MB)
    levels = [L]
    while levels[-1] > 1:
        levels.append(-(-levels[-1] // RAD))
    Q = [dict() for _ in levels]
    R = [dict() for _ in levels]
    for i in range(L):
        blk = np.zeros((MB, NB))
        rows = P[i * MB:(i + 1) * MB]
        blk[:rows.shape[0]] = rows
        R[0][i], Q[0][i] = house_qr_inplace(blk)
    for l in range(1, len(levels)):
        for i in range(levels[l]):
            blk = np.zeros((MB, NB))
            for k in range(RAD):
                c = RAD * i + k
                if c < levels[l - 1]:
                    blk[NB * k:NB * (k + 1)] = R[l - 1][c]
            R[l][i], Q[l][i] = house_qr_inplace(blk)
    top = len(levels) - 1
    Rt = R[top][0]

    def path(i):
        """B(i): 64 x 64 with rows of Q1 owned by leaf i = Q_leaf(i) @ B(i)."""
        B = np.eye(NB)
        idx = i
        for l in range(1, len(levels)):
            k = idx % RAD
            idx //= RAD
            B = B @ Q[l][idx][NB * k:NB * (k + 1), :]
        return B
    # Q11 = top block of Q1 (leftmost path)
    X = Q[0][0][:NB, :] @ path(0)
    sg = np.zeros(NB)
    for j in range(NB):                      # modified LU of Q11 - S', S' chosen on the fly
        d = X[j, j]
        sg[j] = -1.0 if d >= 0 else 1.0
        X[j, j] = d - sg[j]
        X[j + 1:, j] /= X[j, j]
        X[j + 1:, j + 1:] -= np.outer(X[j + 1:, j], X[j, j + 1:])
    Lt = np.tril(X, -1) + np.eye(NB)
    U = np.triu(X)
    Uinv = np.linalg.inv(U)
    T = -(U * sg[None, :]) @ np.linalg.inv(Lt).T
    V = np.zeros((L * MB, NB)); W = np.zeros_like(V)
    for i in range(L):
        BV = path(i) @ Uinv
        V[i * MB:(i + 1) * MB] = Q[0][i] @ BV
        W[i * MB:(i + 1) * MB] = Q[0][i] @ (BV @ T)
    V[:NB] = Lt
    W[:NB] = Lt @ T
    return V[:m], T, W[:m], sg[:, None] * Rt


def check(m, seed, cond=None, zero_cols=0):
    rng = np.random.default_rng(seed)
    P = rng.standard_normal((m, NB))
    if cond:
        u, s, vt = np.linalg.svd(P, full_matrices=False)
        P = (u * np.logspace(0, -cond, NB)) @ vt
    if zero_cols:
        P[:, -zero_cols:] = 0.0
        P[m - 40:, :] = 0.0
    V, T, W, R = tsqr_hr(P)
    Qf = np.eye(m) - V @ T @ V.T
    e_orth = np.max(np.abs(Qf.T @ Qf - np.eye(m)))
    QtP = Qf.T @ P
    e_fact = np.max(np.abs(QtP[:NB] - R)) / np.max(np.abs(P))
    e_zero = np.max(np.abs(QtP[NB:])) / np.max(np.abs(P)) if m > NB else 0.0
    e_struct = max(np.max(np.abs(np.triu(V[:NB], 1))), np.max(np.abs(np.diag(V[:NB]) - 1)), np.max(np.abs(np.tril(T, -1))), np.max(np.abs(np.tril(R, -1))))
    e_w = np.max(np.abs(W - V @ T))
    print("m=%5d cond=%s zero=%d: orth %.1e  Q^T P = [R;0] %.1e / %.1e  structure %.1e  W %.1e  max|V| %.2f max|T| %.2f"
          % (m, cond, zero_cols, e_orth, e_fact, e_zero, e_struct, e_w, np.max(np.abs(V)), np.max(np.abs(T))))
    assert e_orth < 5e-14 and e_fact < 5e-14 and e_zero < 5e-14 and e_struct == 0.0 and e_w < 1e-13


if __name__ == "__main__":
    for m in (64, 128, 192, 256, 320, 1024, 1088, 4032):
        check(m, m)
    check(1024, 1, cond=12)
    check(2048, 2, cond=15)
    check(576, 3, zero_cols=24)
    print("ok")
