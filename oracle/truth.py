"""ctypes front-end of oracle/truth_quad.c (TEST INFRASTRUCTURE ONLY): 113-bit bisection on the LDL^T inertia of a
banded pencil H - x S.  Used by tests/golden/make_truth.py (truth fixtures of the reference's pencils,
matrices.f90:244-248) and by the GPU stage tests (truth of synthetic tridiagonal matrices: k = 2, S = I)."""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libtruth.so")
        src = os.path.join(_HERE, "truth_quad.c")
        if not os.path.exists(so) or (os.path.exists(src) and os.path.getmtime(so) < os.path.getmtime(src)):
            subprocess.check_call(["gcc", "-O2", "-fopenmp", "-fPIC", "-shared", "-o", so, src])
        L = C.CDLL(so)
        L.orc_truth_eigs.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_double, C.c_double, C.c_void_p, C.c_void_p]
        L.orc_truth_count.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_double, C.c_double]
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def band_eigs(SB, HB, idx, est, lam, rtol=1e-24):
    """Eigenvalues number idx (0-based) of the pencil (HB, SB) (upper bands [k][n]) near the estimates est;
    returns (hi, lo) with truth = hi + lo."""
    k, n = SB.shape
    idx = np.ascontiguousarray(idx, dtype=np.int32)
    est = np.ascontiguousarray(est, dtype=np.float64)
    width = np.full(len(idx), 64 * np.finfo(float).eps * lam)
    hi = np.zeros(len(idx)); lo = np.zeros(len(idx))
    SBc = np.ascontiguousarray(SB, dtype=np.float64); HBc = np.ascontiguousarray(HB, dtype=np.float64)
    st = lib().orc_truth_eigs(n, k, _p(SBc), _p(HBc), len(idx), _p(idx), _p(est), _p(width), rtol, 1e-30 * lam,
                              _p(hi), _p(lo))
    if st:
        raise RuntimeError("truth bracket %d never closed" % (st - 1))
    return hi, lo


def tridiag_eigs(d, e, est):
    """All eigenvalues of the symmetric tridiagonal (d, e) near the estimates est (ascending, one per index)."""
    n = len(d)
    SB = np.zeros((2, n)); SB[0] = 1.0
    HB = np.zeros((2, n)); HB[0] = d; HB[1, :n - 1] = e
    lam = max(float(np.max(np.abs(d))) + 2 * float(np.max(np.abs(e))) if n > 1 else float(abs(d[0])), 1e-300)
    hi, lo = band_eigs(SB, HB, np.arange(n), est, lam)
    return hi
