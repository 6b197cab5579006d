"""numpy prototypes of the two-stage tridiagonalisation, written with the same data layout and
index conventions as the HIP kernels (bspatom_amd/csrc/*.hip) so the math can be checked on the CPU.

  sy2sb : dense symmetric (full storage) -> band of half-width nb
          per panel: Householder QR of A[r0:, c0:c0+nb] -> V (explicit unit lower trapezoid), tau, R
                     T from G = V^T V (dlarft forward/columnwise recurrence)
                     W = V T ; Y = A22 W ; K = W^T Y ; Z = Y - 1/2 V K ; A22 -= V Z^T + Z V^T
  sb2st : band (lower, LD = 2*nb rows: AB[d, j] = A[j+d, j]) -> tridiagonal by bulge chasing
"""
import numpy as np


def house(x):
    """LAPACK dlarfg: returns (beta, tau, v) with v[0]=1, (I - tau v v^T) x = beta e1."""
    alpha = x[0]
    xnorm = np.linalg.norm(x[1:])
    if xnorm == 0.0:
        return alpha, 0.0, np.concatenate([[1.0], np.zeros(len(x) - 1)])
    beta = -np.copysign(np.hypot(alpha, xnorm), alpha)
    tau = (beta - alpha) / beta
    v = x / (alpha - beta)
    v[0] = 1.0
    return beta, tau, v


def panel_qr(P):
    """Unblocked Householder QR of m x nb panel. Returns R-in-place panel, explicit V, tau."""
    m, nb = P.shape
    P = P.copy()
    V = np.zeros((m, nb))
    tau = np.zeros(nb)
    for j in range(min(nb, m)):
        beta, t, v = house(P[j:, j].copy())
        tau[j] = t
        V[j:, j] = v
        P[j, j] = beta
        P[j + 1:, j] = 0.0
        if j + 1 < nb:
            w = v @ P[j:, j + 1:]
            P[j:, j + 1:] -= t * np.outer(v, w)
    return P, V, tau


def form_T(V, tau):
    nb = V.shape[1]
    G = V.T @ V
    T = np.zeros((nb, nb))
    for j in range(nb):
        T[j, j] = tau[j]
        if j > 0:
            T[:j, j] = -tau[j] * (T[:j, :j] @ G[:j, j])
    return T


def sy2sb(A, nb):
    A = A.copy()
    n = A.shape[0]
    assert n % nb == 0
    for c0 in range(0, n - nb, nb):
        r0 = c0 + nb
        P, V, tau = panel_qr(A[r0:, c0:c0 + nb])
        A[r0:, c0:c0 + nb] = P
        A[c0:c0 + nb, r0:] = P.T
        T = form_T(V, tau)
        W = V @ T
        A22 = A[r0:, r0:]
        Y = A22 @ W
        K = W.T @ Y
        Z = Y - 0.5 * V @ K
        A22 -= V @ Z.T + Z @ V.T
    return A


def dense_to_band(A, nb):
    n = A.shape[0]
    AB = np.zeros((2 * nb, n))
    for j in range(n):
        for d in range(0, min(nb, n - 1 - j) + 1):
            AB[d, j] = A[j + d, j]
    return AB


def sb2st(AB, b):
    """Bulge chasing on lower band storage AB[d, j] = A[j+d, j], d < 2b.  Works on a dense
    scratch view for clarity; the HIP kernel does the same block operations on AB directly."""
    n = AB.shape[1]
    A = np.zeros((n, n))
    for j in range(n):
        for d in range(min(2 * b, n - j)):
            A[j + d, j] = AB[d, j]
            A[j, j + d] = AB[d, j]
    for s in range(n - 2):
        # task type 1: annihilate A[s+2 : s+1+L, s]
        L = min(b, n - 1 - s)
        if L < 2:
            continue
        r0 = s + 1
        beta, tau, v = house(A[r0:r0 + L, s].copy())
        A[r0, s] = beta; A[r0 + 1:r0 + L, s] = 0.0
        A[s, r0] = beta; A[s, r0 + 1:r0 + L] = 0.0
        D = A[r0:r0 + L, r0:r0 + L]
        p = tau * (D @ v)
        alpha = -0.5 * tau * (p @ v)
        p = p + alpha * v
        D -= np.outer(v, p) + np.outer(p, v)
        while r0 + L < n:
            L2 = min(b, n - (r0 + L))
            B = A[r0 + L:r0 + L + L2, r0:r0 + L]        # L2 x L block below the diagonal block
            # right-apply H (tau, v): B <- B (I - tau v v^T)
            w = B @ v
            B -= tau * np.outer(w, v)
            # new reflector from first column of B
            beta2, tau2, v2 = house(B[:, 0].copy())
            B[0, 0] = beta2; B[1:, 0] = 0.0
            # left-apply to the other columns
            if L > 1:
                z = v2 @ B[:, 1:]
                B[:, 1:] -= tau2 * np.outer(v2, z)
            A[r0:r0 + L, r0 + L:r0 + L + L2] = B.T
            # two-sided on next diagonal block
            D2 = A[r0 + L:r0 + L + L2, r0 + L:r0 + L + L2]
            p = tau2 * (D2 @ v2)
            alpha = -0.5 * tau2 * (p @ v2)
            p = p + alpha * v2
            D2 -= np.outer(v2, p) + np.outer(p, v2)
            r0, L, tau, v = r0 + L, L2, tau2, v2
    d = np.diag(A).copy()
    e = np.diag(A, -1).copy()
    # verify tridiagonal
    off = A - np.diag(d) - np.diag(e, -1) - np.diag(e, 1)
    return d, e, np.max(np.abs(off))


if __name__ == "__main__":
    rng = np.random.default_rng(1)
    for n, nb in [(64, 8), (96, 16), (128, 32), (256, 64)]:
        M = rng.standard_normal((n, n)); M = M + M.T
        ref = np.linalg.eigvalsh(M)
        Bd = sy2sb(M, nb)
        # band structure
        mask = np.abs(np.subtract.outer(np.arange(n), np.arange(n))) > nb
        print(n, nb, "out-of-band max", np.max(np.abs(Bd[mask])), "eig err", np.max(np.abs(np.linalg.eigvalsh(Bd) - ref)))
        AB = dense_to_band(Bd, nb)
        d, e, off = sb2st(AB, nb)
        from scipy.linalg import eigvalsh_tridiagonal
        print("   sb2st off", off, "eig err", np.max(np.abs(eigvalsh_tridiagonal(d, e) - ref)))
