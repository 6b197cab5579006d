"""Debug helper: sb2st (version from BSP_SB2ST_VERSION) against LAPACK on random band matrices of several sizes."""
import sys, numpy as np
sys.path.insert(0, "/root/repo")
from bspatom_amd import capi
from scipy.linalg import eigvalsh_tridiagonal, eig_banded

def run(n, batch=1, seed=0):
    npad = (n + 63) // 64 * 64
    rng = np.random.default_rng(seed)
    AB = np.zeros((batch, npad, 128))
    for b in range(batch):
        for j in range(n):
            m = min(64, n - 1 - j)
            AB[b, j, :m + 1] = rng.standard_normal(m + 1)
    try:
        d, e = capi.stage_sb2st(AB, n)
    except Exception as ex:
        return f"n={n}: {ex}"
    errs = []
    for b in range(batch):
        lower = np.zeros((65, n))
        for j in range(n):
            lower[:, j] = AB[b, j, :65]
        ref = eig_banded(lower, lower=True, eigvals_only=True)
        got = eigvalsh_tridiagonal(d[b], e[b])
        errs.append(np.max(np.abs(ref - got)) / np.max(np.abs(ref)))
    return f"n={n}: max rel eig err {max(errs):.2e}"

if __name__ == "__main__":
    for n in [int(a) for a in sys.argv[1:]] or [66, 100, 128, 129, 130, 131, 160, 192, 193, 194, 200, 250, 256, 320, 700]:
        print(run(n), flush=True)
