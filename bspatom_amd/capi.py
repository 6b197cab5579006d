"""ctypes binding of libbspatom.so (include/bspatom.h).  No torch, no numpy fallbacks: if the
shared library or a gfx950 device is missing the calls fail loudly (there is no CPU path)."""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libbspatom.so")

ERRORS = {-1: "HIP runtime error", -2: "invalid argument", -3: "FATAL ERROR - BSPLVB",
          -4: "no gfx950 device (libbspatom has no CPU path)", -5: "unsupported"}


class BspAtomError(RuntimeError):
    def __init__(self, code, where=""):
        self.code = code
        super().__init__("%s failed: %s (code %d)" % (where or "libbspatom", ERRORS.get(code, "error"), code))


class Input(C.Structure):
    _fields_ = [("kind_grid", C.c_int32), ("k", C.c_int32), ("ka", C.c_int32), ("nfun", C.c_int32),
                ("kind_bc1", C.c_int32), ("kind_bc2", C.c_int32),
                ("ra", C.c_double), ("rb", C.c_double), ("rmax", C.c_double),
                ("n0_ini", C.c_int32), ("l_ini", C.c_int32), ("m_ini", C.c_int32), ("l_fin", C.c_int32),
                ("lmax", C.c_int32), ("kind_pot", C.c_int32),
                ("emax_fin", C.c_double), ("zatom", C.c_double)]


class Sizes(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("nfun", "k", "ka", "nkp", "nointv", "nbc1", "nbc2", "lmax",
                                         "nintv_exp", "nintv_lin", "npad")]


EXPORTS = ["bspatom_input_defaults", "bspatom_device_count", "bspatom_host_setup", "bspatom_problem_create", "bspatom_problem_destroy",
           "bspatom_problem_sizes", "bspatom_problem_grid", "bspatom_problem_route", "bspatom_assemble", "bspatom_solve", "bspatom_solve_dev",
           "bspatom_eigvec", "bspatom_eigvecs", "bspatom_dipole_bands", "bspatom_dipole_elements", "bspatom_write_wf", "bspatom_last_timing", "bspatom_early_vector_state", "bsp_dsygv_", "bspatom_stage_gemm",
           "bspatom_stage_standard_form", "bspatom_stage_sy2sb", "bspatom_stage_panel", "bspatom_stage_sb2st", "bspatom_stage_sb2sb", "bspatom_stage_bisect", "bspatom_stage_crawford", "bspatom_stage_band_eigenvalue",
           "bspatom_release_scratch", "bspatom_run_token", "bspatom_comm_create", "bspatom_comm_allgather", "bspatom_comm_collectives", "bspatom_comm_destroy",
           "bspatom_set_option", "bspatom_get_option", "bspatom_kernel_times", "bspatom_kernel_slot_name"]

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("libbspatom.so is not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(or make -C bspatom_amd/csrc); there is no fallback implementation")
        L = C.CDLL(LIB_PATH)
        vp, i32, dbl, lng = C.c_void_p, C.c_int, C.c_double, C.c_long
        L.bspatom_input_defaults.argtypes = [C.POINTER(Input)]
        L.bspatom_input_defaults.restype = None
        L.bspatom_host_setup.argtypes = [C.POINTER(Input), C.POINTER(Sizes), vp, vp, vp, vp]
        L.bspatom_problem_create.argtypes = [C.POINTER(Input), i32, C.POINTER(vp)]
        L.bspatom_problem_destroy.argtypes = [vp]
        L.bspatom_problem_destroy.restype = None
        L.bspatom_problem_sizes.argtypes = [vp, C.POINTER(Sizes)]
        L.bspatom_problem_grid.argtypes = [vp, vp, vp, vp, vp]
        L.bspatom_problem_route.argtypes = [vp]
        L.bspatom_assemble.argtypes = [vp, i32, i32, vp, vp]
        L.bspatom_solve.argtypes = [vp, i32, i32, vp, vp]
        L.bspatom_solve_dev.argtypes = [vp, i32, i32, vp, vp]
        L.bspatom_eigvec.argtypes = [vp, i32, i32, vp]
        L.bspatom_eigvecs.argtypes = [vp, i32, i32, i32, vp]
        L.bspatom_dipole_elements.argtypes = [vp, i32, i32, i32, i32, i32, vp, vp]
        L.bspatom_dipole_bands.argtypes = [vp, vp]
        L.bspatom_write_wf.argtypes = [vp, vp, i32, vp, vp]
        L.bspatom_last_timing.argtypes = [vp, vp]
        L.bspatom_early_vector_state.argtypes = [vp, vp]
        L.bspatom_stage_gemm.argtypes = [i32, i32, i32, i32, vp, lng, lng, lng, lng, vp, lng, lng, lng, lng,
                                         vp, lng, lng, lng, lng, dbl, dbl]
        L.bspatom_stage_standard_form.argtypes = [i32, i32, i32, vp, vp, vp, vp, vp]
        L.bspatom_stage_sy2sb.argtypes = [i32, i32, vp, vp]
        L.bspatom_stage_sb2st.argtypes = [i32, i32, i32, vp, vp, vp]
        L.bspatom_stage_panel.argtypes = [i32, i32, i32, vp, vp, vp]
        L.bspatom_stage_sb2sb.argtypes = [i32, i32, i32, vp]
        L.bspatom_stage_bisect.argtypes = [i32, i32, vp, vp, vp]
        L.bspatom_stage_crawford.argtypes = [i32, i32, i32, vp, vp, vp, vp]
        L.bspatom_stage_band_eigenvalue.argtypes = [i32, i32, vp, vp, i32, vp]
        L.bsp_dsygv_.restype = None
        L.bspatom_set_option.argtypes = [C.c_char_p, i32]
        L.bspatom_get_option.argtypes = [C.c_char_p, C.POINTER(i32)]
        L.bspatom_kernel_times.argtypes = [vp, vp, i32]
        L.bspatom_kernel_slot_name.argtypes = [i32]
        L.bspatom_kernel_slot_name.restype = C.c_char_p
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _chk(rc, where):
    if rc != 0:
        raise BspAtomError(rc, where)


def make_input(**kw):
    inp = Input()
    lib().bspatom_input_defaults(C.byref(inp))
    names = {f[0] for f in Input._fields_}
    for key, v in kw.items():
        key = key.lower()
        if key in names:
            setattr(inp, key, v)
    return inp


def host_setup(inp, arrays=True):
    """READ_INPUTS sizes and GRID/gauleg arrays, computed on the host (works without a GPU)."""
    s = Sizes()
    _chk(lib().bspatom_host_setup(C.byref(inp), C.byref(s), None, None, None, None), "bspatom_host_setup")
    if not arrays:
        return s
    rt = np.zeros(s.nkp); aind = np.zeros(2 * s.nfun); xg = np.zeros(s.ka); wg = np.zeros(s.ka)
    _chk(lib().bspatom_host_setup(C.byref(inp), C.byref(s), _p(rt), _p(aind), _p(xg), _p(wg)), "bspatom_host_setup")
    return s, rt, aind, xg, wg


class Problem:
    """One B-spline radial problem resident on one MI355X (wraps bspatom_problem)."""

    def __init__(self, inp, device=0):
        self._h = C.c_void_p()
        _chk(lib().bspatom_problem_create(C.byref(inp), device, C.byref(self._h)), "bspatom_problem_create")
        s = Sizes()
        _chk(lib().bspatom_problem_sizes(self._h, C.byref(s)), "bspatom_problem_sizes")
        self.sizes = s
        self.inp = inp
        for n, _ in Sizes._fields_:
            setattr(self, n, getattr(s, n))

    def close(self):
        if self._h:
            lib().bspatom_problem_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def grid(self):
        rt = np.zeros(self.nkp); aind = np.zeros(2 * self.nfun); xg = np.zeros(self.ka); wg = np.zeros(self.ka)
        _chk(lib().bspatom_problem_grid(self._h, _p(rt), _p(aind), _p(xg), _p(wg)), "bspatom_problem_grid")
        return rt, aind, xg, wg

    def assemble(self, l0, nl):
        SB = np.zeros((self.k, self.nfun)); HB = np.zeros((nl, self.k, self.nfun))
        _chk(lib().bspatom_assemble(self._h, l0, nl, _p(SB), _p(HB)), "bspatom_assemble")
        return SB, HB

    def solve(self, l0, nl):
        E = np.zeros((nl, self.nfun)); info = np.zeros(nl, dtype=np.int32)
        _chk(lib().bspatom_solve(self._h, l0, nl, _p(E), _p(info)), "bspatom_solve")
        return E, info

    def solve_dev(self, l0, nl, dev_ptr):
        info = np.zeros(nl, dtype=np.int32)
        _chk(lib().bspatom_solve_dev(self._h, l0, nl, C.c_void_p(dev_ptr), _p(info)), "bspatom_solve_dev")
        return info

    def eigvec(self, l, n0):
        c = np.zeros(self.nfun)
        _chk(lib().bspatom_eigvec(self._h, l, n0, _p(c)), "bspatom_eigvec")
        return c

    def dipole_bands(self):
        """Full bands (3, 2k-1, nfun) of int B_i r B_j, int B_i (1/r) B_j, int B_i B_j' (rij of KIND_PI = 1, 2)."""
        RB = np.zeros((3, 2 * self.k - 1, self.nfun))
        _chk(lib().bspatom_dipole_bands(self._h, _p(RB)), "bspatom_dipole_bands")
        return RB

    def eigvecs(self, l, n0, count):
        """Eigenvectors n0 .. n0+count-1 (1-based) of channel l: array (count, nfun), each S-normalised."""
        Z = np.zeros((count, self.nfun))
        _chk(lib().bspatom_eigvecs(self._h, l, n0, count, _p(Z)), "bspatom_eigvecs")
        return Z

    def dipole_elements(self, l_ini, n0_ini, l_fin, n0_fin, count, a):
        """D[i] = c(l_fin, n0_fin+i)^T (a[0] R_r + a[1] R_1/r + a[2] R_d/dr) c(l_ini, n0_ini), 1-based state numbers:
        the DGEMV + DDOT of TRANS_AMP (PhotoIon.f90:95-107) for channels of the last solved batch."""
        a = np.ascontiguousarray(a, dtype=np.float64)
        assert a.shape == (3,)
        D = np.zeros(count)
        _chk(lib().bspatom_dipole_elements(self._h, l_ini, n0_ini, l_fin, n0_fin, count, _p(a), _p(D)), "bspatom_dipole_elements")
        return D

    def write_wf(self, c, npts=10000):
        c = np.ascontiguousarray(c, dtype=np.float64)
        r = np.zeros(npts + 1); u = np.zeros(npts + 1)
        _chk(lib().bspatom_write_wf(self._h, _p(c), npts, _p(r), _p(u)), "bspatom_write_wf")
        return r, u

    def early_vector_state(self):
        """0: the last solve computed no early vector; 1: computed and kept; -1: computed, failed its check, dropped (include/bspatom.h)."""
        st = C.c_int32(0)
        _chk(lib().bspatom_early_vector_state(self._h, C.byref(st)), "bspatom_early_vector_state")
        return st.value

    def route(self):
        """2 = band route (csrc/crawford.hip), 1 = dense route: what solve() takes under the current switches"""
        r = lib().bspatom_problem_route(self._h)
        _chk(min(r, 0), "bspatom_problem_route")
        return r

    def last_timing(self):
        ms = np.zeros(6)
        _chk(lib().bspatom_last_timing(self._h, _p(ms)), "bspatom_last_timing")
        return dict(zip(("assemble", "chol_std", "sy2sb", "sb2st", "bisect", "total"), ms.tolist()))


# ---- stage-level helpers (parity tests) ------------------------------------------------------
def _flat(x):
    base = x
    while base.base is not None:
        base = base.base
    return base


def stage_gemm(A, B, C_, alpha=1.0, beta=0.0, transA=False, transB=False):
    """C = alpha*op(A)@op(B) + beta*C on the MFMA kernel.  A, B, C_: (batch, rows, cols) numpy views whose
    element strides are passed through unchanged (so both C- and F-ordered matrices can be exercised)."""
    assert A.ndim == 3 and B.ndim == 3 and C_.ndim == 3
    opA = A.transpose(0, 2, 1) if transA else A
    opB = B.transpose(0, 2, 1) if transB else B
    batch, M, K = opA.shape
    N = opB.shape[2]
    es = 8
    fa, fb, fc = _flat(A), _flat(B), _flat(C_)
    rc = lib().bspatom_stage_gemm(M, N, K, batch,
                                  _p(fa), opA.strides[1] // es, opA.strides[2] // es, opA.strides[0] // es, fa.size,
                                  _p(fb), opB.strides[1] // es, opB.strides[2] // es, opB.strides[0] // es, fb.size,
                                  _p(fc), C_.strides[1] // es, C_.strides[2] // es, C_.strides[0] // es, fc.size,
                                  alpha, beta)
    _chk(rc, "bspatom_stage_gemm")
    return C_


def stage_standard_form(SB, HB):
    k, n = SB.shape
    nl = HB.shape[0]
    npad = (n + 63) // 64 * 64
    UB = np.zeros((k, n)); Cm = np.zeros((nl, npad, npad)); info = np.zeros(1, dtype=np.int32)
    _chk(lib().bspatom_stage_standard_form(n, k, nl, _p(np.ascontiguousarray(SB)), _p(np.ascontiguousarray(HB)),
                                           _p(UB), _p(Cm), _p(info)), "bspatom_stage_standard_form")
    # device layout is column-major per channel: C[l][j*npad + i] = C(i,j) -> transpose view
    return UB, Cm.transpose(0, 2, 1), int(info[0])


def stage_sy2sb(A):
    """A: (batch, npad, npad) symmetric.  Returns AB (batch, npad, 128): AB[b, j, d] = band(j+d, j)."""
    batch, npad, _ = A.shape
    Af = np.ascontiguousarray(A.transpose(0, 2, 1))      # column-major per channel
    AB = np.zeros((batch, npad, 128))
    _chk(lib().bspatom_stage_sy2sb(npad, batch, _p(Af), _p(AB)), "bspatom_stage_sy2sb")
    return AB


def stage_panel(A, c0):
    """Panel factorisation of sy2sb alone.  A: (batch, npad, npad), any matrix (only the panel A[:, c0+64:, c0:c0+64] is touched).
    Returns (A with the panel replaced by [R; 0], V (batch, m, 64), W (batch, m, 64))."""
    batch, npad, _ = A.shape
    m = npad - c0 - 64
    Af = np.ascontiguousarray(A.transpose(0, 2, 1))      # column-major per matrix
    V = np.zeros((batch, 64, m)); W = np.zeros((batch, 64, m))
    _chk(lib().bspatom_stage_panel(npad, c0, batch, _p(Af), _p(V), _p(W)), "bspatom_stage_panel")
    return Af.transpose(0, 2, 1), V.transpose(0, 2, 1), W.transpose(0, 2, 1)


def stage_sb2st(AB, n):
    batch, npad, _ = AB.shape
    d = np.zeros((batch, npad)); e = np.zeros((batch, npad))
    _chk(lib().bspatom_stage_sb2st(n, npad, batch, _p(np.ascontiguousarray(AB)), _p(d), _p(e)), "bspatom_stage_sb2st")
    return d[:, :n], e[:, :n - 1]


def stage_sb2sb(AB, n):
    """First half of the two-step route: band 64 -> band 16 (returns the band array, same layout)."""
    batch, npad, _ = AB.shape
    out = np.ascontiguousarray(AB).copy()
    _chk(lib().bspatom_stage_sb2sb(n, npad, batch, _p(out)), "bspatom_stage_sb2sb")
    return out


def stage_crawford(SB, HB):
    """Band route, first stage: upper bands SB (k, n), HB (nl, k, n) -> (AB (nl, npad, 128) with AB[l, j, d] = A_l(j + d, j), info)."""
    nl, k, n = HB.shape
    npad = (n + 63) // 64 * 64
    AB = np.zeros((nl, npad, 128))
    info = C.c_int32(0)
    _chk(lib().bspatom_stage_crawford(n, k, nl, _p(np.ascontiguousarray(SB, dtype=np.float64)),
                                      _p(np.ascontiguousarray(HB, dtype=np.float64)), _p(AB), C.byref(info)), "bspatom_stage_crawford")
    return AB, info.value


def stage_band_eigenvalue(SB, HB, m):
    """Eigenvalue m (0-based, ascending) of the banded pencil (HB, SB) (upper bands (k, n)) from inertia counts (csrc/bandsect.hip)."""
    k, n = HB.shape
    lam = np.zeros(1)
    _chk(lib().bspatom_stage_band_eigenvalue(n, k, _p(np.ascontiguousarray(SB, dtype=np.float64)),
                                             _p(np.ascontiguousarray(HB, dtype=np.float64)), m, _p(lam)), "bspatom_stage_band_eigenvalue")
    return lam[0]


def stage_bisect(d, e):
    batch, n = d.shape
    ee = np.zeros((batch, n)); ee[:, :n - 1] = e
    w = np.zeros((batch, n))
    _chk(lib().bspatom_stage_bisect(n, batch, _p(np.ascontiguousarray(d)), _p(ee), _p(w)), "bspatom_stage_bisect")
    return w


def dsygv(A, B, jobz="V", uplo="U"):
    """bsp_dsygv_ through its Fortran-77 ABI.  A, B: (n,n) arrays; returns w, Z (columns), factor of B, info."""
    n = A.shape[0]
    a = np.array(A, dtype=np.float64, order="F")
    b = np.array(B, dtype=np.float64, order="F")
    w = np.zeros(n); work = np.zeros(max(1, 4 * n))
    it = C.c_int(1); nn = C.c_int(n); lda = C.c_int(n); lw = C.c_int(4 * n); info = C.c_int(0)
    lib().bsp_dsygv_(C.byref(it), C.c_char_p(jobz.encode()), C.c_char_p(uplo.encode()), C.byref(nn), _p(a), C.byref(lda),
                     _p(b), C.byref(lda), _p(w), _p(work), C.byref(lw), C.byref(info), C.c_size_t(1), C.c_size_t(1))
    return w, a, b, info.value


def set_option(name, value):
    """Flip one of the library's run-time switches (BSP_* variables of DESIGN.md 4.4) in this process."""
    _chk(lib().bspatom_set_option(name.encode(), int(value)), "bspatom_set_option(%s)" % name)


def kernel_times():
    """Launch durations recorded since the last call while option "ktime" was 1: {slot name: (sum of ms, launches)}."""
    ms = np.zeros(16); cnt = np.zeros(16, dtype=np.int32)
    ns = lib().bspatom_kernel_times(_p(ms), _p(cnt), 16)
    if ns < 0:
        raise BspAtomError(ns, "bspatom_kernel_times")
    return {lib().bspatom_kernel_slot_name(i).decode(): (float(ms[i]), int(cnt[i])) for i in range(ns)}


def get_option(name):
    v = C.c_int(0)
    _chk(lib().bspatom_get_option(name.encode(), C.byref(v)), "bspatom_get_option(%s)" % name)
    return v.value
