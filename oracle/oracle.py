"""ctypes front-end of oracle/bsp_oracle.c (TEST INFRASTRUCTURE ONLY).

Parity status: PINNED -- checked in tests/test_oracle_golden.py against fixtures produced
by the compiled reference (tests/golden/make_golden.py): derived sizes, knots, Gauss-Legendre
rule and Aind bit-for-bit; S, T, V bit-for-bit; U_l bit-for-bit; spectra within the stated
LAPACK-vs-LAPACK tolerance (the reference's DSYGV lives in Intel MKL, version unpinned --
matrices.f90:248, src/Makefile:23; the oracle uses netlib LAPACK 3.12 DSYGV from scipy's
OpenBLAS, the same library the compiled reference in oracle/_ref is linked to, or the plain
C chain orc_dsygv for small sizes).
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class OrcCfg(C.Structure):
    _fields_ = [
        ("kind_grid", C.c_int), ("k", C.c_int), ("ka", C.c_int), ("nfun", C.c_int),
        ("kind_bc1", C.c_int), ("kind_bc2", C.c_int),
        ("ra", C.c_double), ("rb", C.c_double), ("rmax", C.c_double),
        ("n0_ini", C.c_int), ("l_ini", C.c_int), ("m_ini", C.c_int), ("l_fin", C.c_int),
        ("lmax", C.c_int), ("kind_pot", C.c_int),
        ("emax_fin", C.c_double), ("zatom", C.c_double),
        ("nbc1", C.c_int), ("nbc2", C.c_int), ("nkp", C.c_int), ("nointv", C.c_int),
        ("nintv_exp", C.c_int), ("nintv_lin", C.c_int), ("imax", C.c_int),
        ("gsize", C.c_double),
        ("numn", C.c_int * 3), ("ntot", C.c_int),
        ("alphan", C.c_double * 3),
        ("bl", C.c_double * 4),
    ]


def build(force=False):
    so = os.path.join(_HERE, "liborc.so")
    src = os.path.join(_HERE, "bsp_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-fPIC", "-shared",
                               "-o", so, src, "-lm"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.orc_selpot.restype = C.c_double
        _LIB.orc_selpot.argtypes = [C.POINTER(OrcCfg), C.c_double]
        _LIB.orc_gauleg.argtypes = [C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_int]
        _LIB.orc_interv.argtypes = [C.c_void_p, C.c_int, C.c_double, C.POINTER(C.c_int)]
        _LIB.orc_bsplvb.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_int, C.c_void_p]
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


# VARS_BSP / VARS_TISE defaults, ReadInputs.f90:27-36, :75-84
DEFAULTS = dict(kind_grid=0, ra=0.0, rb=0.0, rmax=0.0, k=0, ka=0, nfun=0, kind_bc1=0, kind_bc2=0,
                n0_ini=1, l_ini=0, m_ini=0, l_fin=0, lmax=0, emax_fin=-1.0, zatom=1.0, kind_pot=0)


def make_cfg(**kw):
    """Build an OrcCfg from namelist values (lower-case keys) and derive sizes."""
    vals = dict(DEFAULTS)
    for key, v in kw.items():
        key = key.lower()
        if key in vals:
            vals[key] = v
    c = OrcCfg()
    for key, v in vals.items():
        setattr(c, key, v)
    lib().orc_derive(C.byref(c))
    return c


def grid(c):
    rt = np.zeros(c.nkp)
    aind = np.zeros(2 * c.nfun)
    lib().orc_grid(C.byref(c), _p(rt), _p(aind))
    xg = np.zeros(c.ka)
    wg = np.zeros(c.ka)
    lib().orc_gauleg(-1.0, 1.0, _p(xg), _p(wg), c.ka)
    return rt, aind, xg, wg


def matrix_svt(c, rt, aind, xg, wg):
    """Reference-faithful dense assembly: returns S, V, T (n,n) and U (lmax+1,n,n), [i,j] indexed."""
    n = c.nfun
    S = np.zeros(n * n); V = np.zeros(n * n); T = np.zeros(n * n); U = np.zeros(n * n * (c.lmax + 1))
    st = lib().orc_matrix_svt(C.byref(c), _p(rt), _p(aind), _p(xg), _p(wg), _p(S), _p(V), _p(T), _p(U))
    if st:
        raise RuntimeError("oracle: FATAL ERROR - BSPLVB (status %d)" % st)
    f = lambda a: a.reshape(n, n).T.copy()
    return f(S), f(V), f(T), np.stack([f(U[l * n * n:(l + 1) * n * n]) for l in range(c.lmax + 1)])


def assemble_bands(c, rt, aind, xg, wg, l0=0, nl=None):
    """Upper bands: SB[d,i] = S(i,i+d); HB[l,d,i] = ((T+U_l)+V)(i,i+d)  (matrices.f90:244)."""
    if nl is None:
        nl = c.lmax + 1 - l0
    n, k = c.nfun, c.k
    SB = np.zeros((k, n)); HB = np.zeros((nl, k, n))
    st = lib().orc_assemble_bands(C.byref(c), _p(rt), _p(aind), _p(xg), _p(wg), l0, nl, _p(SB), _p(HB))
    if st:
        raise RuntimeError("oracle: FATAL ERROR - BSPLVB (status %d)" % st)
    return SB, HB


def dipole_bands(c, rt, aind, xg, wg):
    """Full bands [3][2k-1][nfun] (RB[c][d+k-1][i] = X_c(i, i+d)) of int B_i r B_j, int B_i (1/r) B_j,
    int B_i B_j' as MATRIX_SVT accumulates them (matrices.f90:141-144; both triangles, not bit-symmetric)."""
    RB = np.zeros((3, 2 * c.k - 1, c.nfun))
    lib().orc_dipole_bands.argtypes = [C.c_void_p] * 6
    st = lib().orc_dipole_bands(C.byref(c), _p(rt), _p(aind), _p(xg), _p(wg), _p(RB))
    if st:
        raise RuntimeError("oracle: FATAL ERROR - BSPLVB (status %d)" % st)
    return RB


def band_to_dense_upper(B):
    k, n = B.shape
    M = np.zeros((n, n))
    for d in range(k):
        idx = np.arange(n - d)
        M[idx, idx + d] = B[d, : n - d]
    return M


def dsygv(H_upper, S_upper, vectors=True, impl="lapack"):
    """DSYGV(1,'V','U') on dense matrices whose upper triangles hold H and S (matrices.f90:248).

    impl='lapack': netlib LAPACK 3.12 through scipy (the published algorithm of the reference's
    third-party dependency); impl='c': the plain C chain in bsp_oracle.c (small n)."""
    n = H_upper.shape[0]
    if impl == "c":
        A = np.asfortranarray(np.triu(H_upper) + np.triu(H_upper, 1).T).copy(order="F")
        B = np.asfortranarray(np.triu(S_upper) + np.triu(S_upper, 1).T).copy(order="F")
        w = np.zeros(n)
        lib().orc_dsygv.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        info = lib().orc_dsygv(n, _p(A), _p(B), _p(w))
        return w, A, info
    return _dsygv_scipy(H_upper, S_upper, vectors)


def _dsygv_scipy(H_upper, S_upper, vectors):
    from scipy.linalg import lapack
    res = lapack.dsygv(np.asfortranarray(H_upper), np.asfortranarray(S_upper), itype=1,
                       jobz="V" if vectors else "N", uplo="U")
    # scipy returns (w, v/a, info) for dsygv
    w, v, info = res[0], res[1], res[-1]
    return w, v, info


def write_wf(c, rt, ci, npts=10000):
    r = np.zeros(npts + 1); u = np.zeros(npts + 1)
    ci = np.ascontiguousarray(ci, dtype=np.float64)
    st = lib().orc_write_wf(C.byref(c), _p(rt), _p(ci), len(ci), npts, _p(r), _p(u))
    if st:
        raise RuntimeError("FATAL ERROR - BSPLVB")
    return r, u


def solve_all(c, vectors_for=None):
    """Whole path on the CPU: returns E[lmax+1, nfun] (and the eigenvector of (l_ini, n0_ini))."""
    rt, aind, xg, wg = grid(c)
    SB, HB = assemble_bands(c, rt, aind, xg, wg)
    S = band_to_dense_upper(SB)
    E = np.zeros((c.lmax + 1, c.nfun))
    vec = None
    for l in range(c.lmax + 1):
        want_v = (l == c.l_ini)
        w, v, info = dsygv(band_to_dense_upper(HB[l]), S, vectors=want_v)
        if info != 0:
            raise RuntimeError("ERROR DIAGONALIZING THE MATRIX! %d (l = %d)" % (info, l))
        E[l] = w
        if want_v:
            vec = v[:, c.n0_ini - 1].copy()
    return E, vec, (rt, aind, xg, wg)


# ---- transition amplitudes of the one-photon branches (KIND_PI = 1, 2) -------------------------------
def three_j(j1, j2, j3, m1, m2, m3):
    """Wigner 3j symbol for integer arguments by the Racah sum over log-factorials, the way the reference's THREE_J
    does it (Funs_WignerSymbols.for:1-62): table FAC(i) = log((i-1)!) built by successive additions, the common
    smallest exponent GROS taken out of the alternating sum, zero unless m1+m2+m3 = 0 and the z range is not empty."""
    import math
    nmax = 140
    fac = [0.0] * (nmax + 1)                       # fac[i] = FAC(i), 1-based
    for i in range(2, nmax + 1):
        fac[i] = fac[i - 1] + math.log(float(i - 1))
    l4 = j1 + j2 + j3 + 2
    if l4 > nmax:
        raise ValueError("THREE_J: arguments beyond the factorial table")
    if m1 + m2 + m3 != 0:
        return 0.0
    izmax = min(j1 + j2 - j3, j1 - m1, j2 + m2) + 1
    izmin = max(0, j2 - j3 - m1, j1 + m2 - j3) + 1
    if izmax - izmin < 0:
        return 0.0
    l1 = j1 + j2 - j3 + 1; l2 = j3 + j1 - j2 + 1; l3 = j3 + j2 - j1 + 1
    l5 = j1 + m1 + 1; l6 = j1 - m1 + 1; l7 = j2 + m2 + 1; l8 = j2 - m2 + 1; l9 = j3 + m3 + 1; l10 = j3 - m3 + 1
    abra = 0.5 * (fac[l1] + fac[l2] + fac[l3] - fac[l4] + fac[l5] + fac[l6] + fac[l7] + fac[l8] + fac[l9] + fac[l10])
    k1 = j3 - j2 + m1 + 1; k2 = j3 - j1 - m2 + 1
    gros = 250.0
    ac = {}
    for ii in range(izmin, izmax + 1):
        i = ii - 1
        ac[ii] = fac[i + 1] + fac[l1 - i] + fac[l6 - i] + fac[l7 - i] + fac[k1 + i] + fac[k2 + i]
        if ac[ii] < gros:
            gros = ac[ii]
    accu = 0.0
    sig = (-1.0) ** izmin
    for ii in range(izmin, izmax + 1):
        sig = -sig
        accu += sig * math.exp(-(ac[ii] - gros))
    return (-1.0) ** (j1 - j2 - m3) * math.exp(abra - gros) * accu


def final_state_limits(E_fin, emax_fin):
    """n0_fin, n1_fin (1-based) as SOLVE_SYSTEM sets them for KIND_PI = 1, 2 at l = l_fin (matrices.f90:272-283):
    last index with E < 0, plus one, capped at nfun-1; last index with E <= Emax_fin (Emax_fin = -1 means En(nfun))."""
    n = len(E_fin)
    if emax_fin == -1.0:
        emax_fin = float(E_fin[-1])
    n0 = -1; n1 = -1
    for i in range(1, n + 1):
        if E_fin[i - 1] < 0.0:
            n0 = i
        if E_fin[i - 1] <= emax_fin:
            n1 = i
    n0 = min(n0 + 1, n - 1)
    return n0, n1, emax_fin


def trans_amp(kind_pi, l0, m0, lf, mf, mph, R1, R2, ci_ini, ci_fin, E_fin, n0_fin, n1_fin):
    """T_fi(ni), ni = n0_fin..n1_fin, of TRANS_AMP for the plane-wave branches (PhotoIon.f90:50-107).  R1, R2: dense
    rij(:,:,1), rij(:,:,2) as MATRIX_SVT leaves them for this KIND_PI; ci_fin[:, ni - n0_fin]."""
    t3a = three_j(lf, 1, l0, -mf, mph, m0)
    if kind_pi == 1:
        t3b = three_j(lf, 1, l0, 0, 0, 0)
        c1 = (-1.0) ** (lf + l0 + mf) * np.sqrt(float((2 * lf + 1) * (2 * l0 + 1))) * t3a * t3b
        c0 = 1.0
        A = c1 * R1
    else:
        c0 = np.sqrt(float(l0 + 1)) * t3a
        c1 = 0.0; c2 = 0.0
        if lf == l0 + 1:
            c1 = float(l0 + 1); c2 = -1.0
        elif lf == l0 - 1:
            c1 = float(l0); c2 = 1.0
        A = c1 * R1 + c2 * R2
    v = A @ ci_ini
    T = np.zeros(n1_fin - n0_fin + 1)
    for ni in range(n0_fin, n1_fin + 1):
        An = np.sqrt(2.0 / (E_fin[ni] - E_fin[ni - 2]))          # E_fin(ni+1) - E_fin(ni-1), 1-based
        T[ni - n0_fin] = An * c0 * float(ci_fin[:, ni - n0_fin] @ v)
    return T
