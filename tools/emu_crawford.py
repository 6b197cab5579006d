#!/usr/bin/env python3
"""emu_crawford.py -- lane-level NumPy emulation of csrc/crawford.hip (the item kernel's register layouts, DPP row sums,
bpermute, the operand maps of v_mfma_f64_16x16x4, the wavefront schedule and the set-up kernels), to check the kernel's index
logic on the CPU before it meets the GPU.  python tools/emu_crawford.py [n k]"""
import sys
import numpy as np
import scipy.linalg as sl

CB = 8
lane = np.arange(64); g = lane >> 4; c = lane & 15; c8 = c & 7; left = c < 8


def rowsum(x):                       # all lanes of a DPP row end with the row's sum
    return x.reshape(4, 16).sum(axis=1).repeat(16)


def bperm(x, src):
    return x[src]


def mfma(a, b, acc):                 # acc: (4, 64): reg r of lane l = D[(l>>4) + 4 r][l & 15]
    A = np.zeros((16, 4)); B = np.zeros((4, 16))
    A[c, g] = a; B[g, c] = b
    Dm = A @ B
    out = acc.copy()
    for r in range(4):
        out[r] += Dm[g + 4 * r, c]
    return out


def rq_step(I, x0, x1, q):
    LEN, GI = CB + I, I & 3
    xr = x0 if I < 4 else x1
    xi = bperm(xr, GI * 16 + c)
    sig = rowsum(np.where(c < LEN, xi * xi, 0.0))
    alpha = xi[LEN]
    a2s = alpha * alpha + sig
    ok = (a2s > 1e-280) & (sig != 0.0)
    nrm = np.sqrt(np.where(ok, a2s, 1.0))
    bt = np.where(alpha >= 0, -nrm, nrm)
    amb = alpha - bt
    with np.errstate(all="ignore"):
        t = np.where(ok, (bt - alpha) / bt, 0.0)
        sc = np.where(ok, 1.0 / amb, 0.0)
    beta = np.where(ok, bt, alpha)
    u = np.where(c < LEN, xi * sc, np.where(c == LEN, 1.0, 0.0))
    if I > 0:
        w0 = rowsum(x0 * u); x0 = x0 - t * w0 * u
    if I > 4:
        w1 = rowsum(x1 * u); x1 = x1 - t * w1 * u
    xr = x0 if I < 4 else x1
    fixed = np.where(c < LEN, 0.0, np.where(c == LEN, beta, xr))
    xr = np.where(g == GI, fixed, xr)
    if I < 4: x0 = xr
    else: x1 = xr
    for r in range(4):
        wq = rowsum(q[r] * u); q[r] = q[r] - t * wq * u
    return x0, x1, q


def rq4(X):
    """crawford_item4_kernel's phase A for one item: X (8 x 16) -> (R part of X, Q 16 x 16), sums as four partial sums"""
    x = np.zeros((16, 16)); x[:8] = X
    q = np.eye(16)

    def dot4(a, b, n):
        ws = [0.0, 0.0, 0.0, 0.0]
        for cc in range(n):
            ws[cc & 3] = a[cc] * b[cc] + ws[cc & 3]
        return (ws[0] + ws[1]) + (ws[2] + ws[3])
    for I in range(7, -1, -1):
        LEN = CB + I
        u = x[I].copy()
        sig = dot4(u, u, LEN); alpha = u[LEN]
        a2s = alpha * alpha + sig
        ok = a2s > 1e-280 and sig != 0.0
        nrm = np.sqrt(a2s if ok else 1.0)
        bt = -nrm if alpha >= 0 else nrm
        amb = alpha - bt
        sc = 1.0 / amb if ok else 0.0
        tt = (bt - alpha) / bt if ok else 0.0
        beta = bt if ok else alpha
        u[:LEN] *= sc; u[LEN] = 1.0; u[LEN + 1:] = 0.0
        for r in range(16):
            w = dot4(x[r], u, LEN + 1)
            tw = -tt * w if r <= I else 0.0
            x[r, :LEN + 1] += tw * u[:LEN + 1]
            w = dot4(q[r], u, LEN + 1)
            q[r, :LEN + 1] += -tt * w * u[:LEN + 1]
    return x[:8], q


ITEMS4 = True


def item(N, t, idx, jlo, nch, jel, Qel, D, E, G):
    if idx < nch:
        elim, j = False, jlo + idx; p = j - 2 - (t - 2 * j)
    elif idx == nch and jel > 0:
        elim, j = True, jel; p = j - 1
    else:
        return
    D0, D1, E0 = D[p], D[p + 1], E[p]
    w = np.zeros((4, 64)); sd = np.zeros((2, 64))
    for r in range(2):
        R = 4 * r + g
        hi = np.maximum(R, c8); lo = np.minimum(R, c8)
        w[r] = np.where(left, D0[hi * CB + lo], E0[c8 * CB + R])
        w[r + 2] = np.where(left, E0[R * CB + c8], D1[hi * CB + lo])
    side = p >= 1
    Em = E[p - 1] if side else E[0]
    for r in range(2):
        sd[r] = np.where(side & left, Em[(4 * r + g) * CB + c8], 0.0)
    q = np.zeros((4, 64)); xt = np.zeros((2, 64))
    has_x = elim and (j + 1 <= N - 1)
    if not elim and ITEMS4:
        X = np.hstack([G[p].reshape(8, 8), E[p + 1].reshape(8, 8)])
        Xr, Qm = rq4(X)
        x0 = Xr[g, c]; x1 = Xr[4 + g, c]
        for r in range(4):
            q[r] = Qm[4 * r + g, c]
    elif not elim:
        src = np.where(left, G[p][(g * CB + c8)], E[p + 1][g * CB + c8])
        x0 = src
        x1 = np.where(left, G[p][((4 + g) * CB + c8)], E[p + 1][(4 + g) * CB + c8])
        for r in range(4):
            q[r] = (4 * r + g == c).astype(float)
        for I in range(7, -1, -1):
            x0, x1, q = rq_step(I, x0, x1, q)
    else:
        for r in range(4):
            q[r] = Qel[j][(4 * r + g) * 16 + c]
        Ej = E[j] if has_x else E[0]
        for r in range(2):
            xt[r] = np.where(has_x & left, Ej[c8 * CB + 4 * r + g], 0.0)
    P = np.zeros((4, 64)); Wn = np.zeros((4, 64)); O = np.zeros((4, 64))
    for r in range(4): P = mfma(w[r], q[r], P)
    for r in range(4): Wn = mfma(q[r], P[r], Wn)
    O = mfma(q[0], sd[0], O); O = mfma(q[1], sd[1], O)
    for r in range(2):
        idxs = (4 * r + g) * CB + c8
        D0[idxs[left]] = Wn[r][left]
        E0[idxs[left]] = Wn[r + 2][left]
        D1[idxs[~left]] = Wn[r + 2][~left]
    if side:
        for r in range(2):
            idxs = (4 * r + g) * CB + c8
            if not elim: E[p - 1][idxs[left]] = O[r][left]
            G[p - 1][idxs[left]] = O[r + 2][left]
    if not elim:
        E1 = E[p + 1]
        E1[(g * CB + c8)[~left]] = x0[~left]
        E1[((4 + g) * CB + c8)[~left]] = x1[~left]
    elif has_x:
        T = np.zeros((4, 64))
        T = mfma(q[2], xt[0], T); T = mfma(q[3], xt[1], T)
        for r in range(2):
            E[j][(c8 * CB + 4 * r + g)[left]] = T[r + 2][left]


def run(SB, HB):
    k, n = SB.shape; b = k - 1
    N = (n + CB - 1) // CB
    SBf = np.zeros_like(SB)
    for d in range(k):
        for i in range(n - d):
            SBf[d, i] = SB[d, n - 1 - i - d]
    Sf = np.zeros((n, n))
    for d in range(k):
        i = np.arange(n - d); Sf[i, i + d] = SBf[d, :n - d]; Sf[i + d, i] = SBf[d, :n - d]
    U = np.linalg.cholesky(Sf).T
    UBf = np.zeros((k, n))
    for d in range(k):
        i = np.arange(n - d); UBf[d, :n - d] = U[i, i + d]
    Qel = np.zeros((N, 256)); LiB = np.zeros((N, 64))
    for j in range(N):
        Ld = np.zeros((8, 8)); M = np.zeros((8, 8))
        for r in range(8):
            for cc in range(8):
                i = CB * j + r; ic = CB * j + cc; d = r - cc
                if i >= n: Ld[r, cc] = 1.0 if r == cc else 0.0
                elif 0 <= d <= b: Ld[r, cc] = UBf[d, ic]
                im = CB * (j - 1) + cc; dm = CB + r - cc
                if j > 0 and i < n and dm <= b: M[r, cc] = UBf[dm, im]
        Li = sl.solve_triangular(Ld, np.eye(8), lower=True)
        K = Li @ M
        LiB[j] = Li.reshape(-1)
        Q = np.zeros((16, 16)); Q[:8, :8] = np.eye(8); Q[:8, 8:] = -K.T; Q[8:, 8:] = Li.T
        Qel[j] = Q.reshape(-1)

    def Hf(i, i2):
        hi = max(i, i2); d = hi - min(i, i2)
        return HB[d, n - 1 - hi] if (hi < n and d <= b) else 0.0
    D = np.zeros((N, 64)); E = np.zeros((N, 64)); G = np.full((N, 64), np.nan)
    for p in range(N):
        for r in range(8):
            for cc in range(8):
                D[p, r * 8 + cc] = Hf(CB * p + r, CB * p + cc); E[p, r * 8 + cc] = Hf(CB * (p + 1) + r, CB * p + cc)
    Li0 = LiB[0].reshape(8, 8)
    D[0] = (Li0 @ D[0].reshape(8, 8) @ Li0.T).reshape(-1); E[0] = (E[0].reshape(8, 8) @ Li0.T).reshape(-1)
    tmax = 3 * N - 5 if N >= 3 else (1 if N == 2 else 0)
    for t in range(1, tmax + 1):
        jel = (t + 1) // 2 if (t & 1) and (t + 1) // 2 <= N - 1 else 0
        jlo = (t + 4) // 3; jhi = min(t // 2, N - 1)
        nch = jhi - jlo + 1 if (jhi >= jlo and jlo >= 2) else 0
        for idx in range(nch + (1 if jel else 0)):
            item(N, t, idx, jlo, nch, jel, Qel, D, E, G)
    A = np.zeros((n, n))                         # band in the ORIGINAL order, as crawford_band_kernel writes it
    for jj in range(n):
        for d in range(16):
            if jj + d < n:
                ihi = n - 1 - jj; ilo = ihi - d; P = ihi >> 3; Pc = ilo >> 3
                blk = D[Pc] if P == Pc else E[Pc]
                v = blk[(ihi & 7) * 8 + (ilo & 7)] if P - Pc <= 1 else 0.0
                A[jj + d, jj] = v; A[jj, jj + d] = v
    return A


if __name__ == "__main__":
    n, k = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (40, 9)
    rng = np.random.default_rng(n)
    SB = np.zeros((k, n)); HB = np.zeros((k, n))
    SB[0] = 2.0 * k + rng.random(n)
    for d in range(1, k): SB[d, :n - d] = rng.standard_normal(n - d)
    for d in range(k): HB[d, :n - d] = rng.standard_normal(n - d)
    A = run(SB, HB)

    def dense(B):
        M = np.zeros((n, n))
        for d in range(k):
            i = np.arange(n - d); M[i, i + d] = B[d, :n - d]; M[i + d, i] = B[d, :n - d]
        return M
    ref = sl.eigh(dense(HB), dense(SB), eigvals_only=True)
    ev = np.linalg.eigvalsh(A)
    print("n %d k %d: emulated kernel vs scipy: %.2e of |lambda|_max" % (n, k, np.max(np.abs(ev - ref)) / np.max(np.abs(ref))))
