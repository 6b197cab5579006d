"""Checks the on-chip forwarding algebra of sb2st v7 on the CPU: sweep s+1's item-k tiles are assembled ONLY from
sweep s's updated tiles of items k and k+1 (plus zeros), never from the band matrix."""
import numpy as np
from proto_twostage import house

def sweep_items(n, b, s):
    """(r0, L, L2) of item 0 (B empty) and the chase items of sweep s."""
    L0 = min(b, n - 1 - s)
    items = [(s + 1, 0, L0)]
    r0, L = s + 1, L0
    while r0 + L < n:
        L2 = min(b, n - (r0 + L))
        items.append((r0, L, L2))
        r0, L = r0 + L, L2
    return items

def run_sweep_tiles(A, b, s, tiles_in=None):
    """Runs sweep s on dense A (in place) item by item; returns the list of updated (B, D2) tiles per item.
    If tiles_in is given (list of (B, D2) per item) the item's inputs are taken from it instead of A (and checked)."""
    n = A.shape[0]
    out = []
    v = None; tau = 0.0
    for k, (r0, L, L2) in enumerate(sweep_items(n, b, s)):
        rn = r0 + L
        Bt = A[rn:rn + L2, r0:r0 + L].copy()
        Dt = A[rn:rn + L2, rn:rn + L2].copy()
        if tiles_in is not None:
            Bi, Di = tiles_in[k][0], tiles_in[k][1]
            assert np.array_equal(Bi, Bt), ("B mismatch", s, k)
            assert np.array_equal(np.tril(Di), np.tril(Dt)), ("D mismatch", s, k)
        if k == 0:
            x = A[s + 1:s + 1 + L2, s].copy()
            if tiles_in is not None:
                assert np.array_equal(tiles_in[k][2], x), ("x mismatch", s)
            beta, tau2, v2 = house(x)
            A[s + 1, s] = beta; A[s + 2:s + 1 + L2, s] = 0.0; A[s, s + 1] = beta; A[s, s + 2:s + 1 + L2] = 0.0
        else:
            w = Bt @ v
            Bt -= tau * np.outer(w, v)
            beta, tau2, v2 = house(Bt[:, 0].copy())
            Bt[0, 0] = beta; Bt[1:, 0] = 0.0
            if L > 1:
                z = v2 @ Bt[:, 1:]
                Bt[:, 1:] -= tau2 * np.outer(v2, z)
        p = tau2 * (Dt @ v2); alpha = -0.5 * tau2 * (p @ v2); p = p + alpha * v2
        Dt -= np.outer(v2, p) + np.outer(p, v2)
        A[rn:rn + L2, r0:r0 + L] = Bt; A[r0:r0 + L, rn:rn + L2] = Bt.T
        A[rn:rn + L2, rn:rn + L2] = Dt
        out.append((Bt.copy(), Dt.copy(), (r0, L, L2)))
        v, tau = v2, tau2
    return out

def assemble_next(tilesA, n, b, s):
    """Inputs of sweep s+1 from sweep s's updated tiles only (the v7 src(a,c) function)."""
    def src(k, a, c):
        Bk, Dk, (r0, L, L2) = tilesA[k]
        R, C = r0 + L, r0
        if c < R:
            ia, jc = a - R, c - C
            return Bk[ia, jc] if (0 <= ia < L2 and 0 <= jc < L) else 0.0
        jc, ia = c - R, a - R
        if ia < L2:
            return Dk[ia, jc]
        if ia == L2 and k + 1 < len(tilesA):
            Bn, Dn, _ = tilesA[k + 1]
            return Bn[0, jc] if jc < L2 else Dn[0, 0]
        return 0.0
    res = []
    for k, (r0, L, L2) in enumerate(sweep_items(n, b, s + 1)):
        rn = r0 + L
        Bi = np.array([[src(k, rn + i, r0 + j) for j in range(L)] for i in range(L2)]).reshape(L2, L)
        Di = np.array([[src(k, rn + i, rn + j) if i >= j else 0.0 for j in range(L2)] for i in range(L2)])
        if k == 0:
            x = np.array([src(0, s + 2 + i, s + 1) for i in range(L2)])
            res.append((Bi, Di, x))
        else:
            res.append((Bi, Di))
    return res

if __name__ == "__main__":
    rng = np.random.default_rng(3)
    for n, b in [(40, 8), (67, 8), (130, 16)]:
        M = np.zeros((n, n))
        for i in range(n):
            for j in range(max(0, i - b), i + 1):
                M[i, j] = M[j, i] = rng.standard_normal()
        ref = np.linalg.eigvalsh(M)
        A = M.copy()
        s = 0
        while s < n - 2 and min(b, n - 1 - s) >= 2:
            tA = run_sweep_tiles(A, b, s)
            if s + 1 < n - 2 and min(b, n - 2 - s) >= 2:
                tin = assemble_next(tA, n, b, s)
                run_sweep_tiles(A, b, s + 1, tiles_in=tin)     # asserts that forwarding reproduces the band exactly
            s += 2
        d = np.diag(A); e = np.diag(A, -1)
        from scipy.linalg import eigvalsh_tridiagonal
        print(n, b, "forwarding exact; eig err", np.max(np.abs(eigvalsh_tridiagonal(d, e) - ref)))
