"""Python host of libbspatom mirroring the reference driver up to the end of SOLVE_SYSTEM
(PROGRAM BSP_ATOM_PI, src/Bsp_Atom.f90:45-95; SOLVE_SYSTEM, src/matrices.f90:204-386):
namelist text -> spectra, the (l_ini, n0_ini) eigenvector, Enl.dat / wf_n0.dat / stdout text in the
reference's formats for KIND_PI = 0, and for KIND_PI >= 3 also the state limits (`select_states`) and
Eigenvec_All.dat.  All arithmetic on matrices happens in libbspatom on the GPU; this module only parses,
dispatches, does the reference's integer bookkeeping on the spectra and formats."""
import math
import os
from . import capi
from .namelist import read_namelists


def fortran_g(v, w, d, e=2):
    """Fortran Gw.d (e = 2) or Gw.dEe edit descriptor (F2008 10.7.5.2.2) for a real value, as gfortran/flang print it:
    F editing with e + 2 trailing blanks inside the range, else 0.dddE+xx with e exponent digits."""
    pad = " " * (e + 2)
    if v == 0.0:
        body = "%.*f" % (d - 1, 0.0)
        return (body + pad).rjust(w)
    return _fortran_g_tail(v, w, d, e, pad, abs(v))


def _fortran_g_tail(v, w, d, edig, pad, a):
    e = math.floor(math.log10(a)) + 1
    if float("%.*e" % (d - 1, a)) >= 10.0 ** e:
        e += 1
    if 0 <= e <= d:
        body = "%.*f" % (d - e, v)
        return (body + pad).rjust(w)
    m = "%.*E" % (d - 1, v)            # d.ddddE+xx -> 0.ddddd E+(xx+1)
    mant, ex = m.split("E")
    sign = "-" if mant.startswith("-") else ""
    digits = mant.replace("-", "").replace(".", "")
    ex = int(ex) + 1
    return ("%s0.%sE%+0*d" % (sign, digits, edig + 1, ex)).rjust(w)


def kind_pi_from_namelist(text):
    return int(read_namelists(text)["vars_field"].get("kind_pi", 0))


def input_from_namelist(text):
    nl = read_namelists(text)
    kw = {}
    kw.update(nl["vars_bsp"]); kw.update(nl["vars_tise"])
    return capi.make_input(**kw)


class StateLimits:
    """What the KIND_PI >= 3 branch of SOLVE_SYSTEM leaves behind: n01[l] = (n0_fin, n1_fin, nE0-1) (1-based, as
    stored), nbold[l], ntemp[l] (columns of Hij kept per channel), n1_max, nbds, emax_fin (as modified), reki."""


def select_states(E, emax_fin, kind_pi):
    """Integer bookkeeping of SOLVE_SYSTEM for KIND_PI >= 3 on the spectra E[l][i] (matrices.f90:290-341 per
    channel, :355-358 for n1_max), statement by statement, including what the reference carries over from one
    channel to the next: `Emax_fin = -1` is replaced ONCE by En(nfun) of l = 0 and later channels then use
    `Elim = Emax_fin + 0.25` (:292-298); n0_fin / n1_fin / ntemp keep the previous channel's value when no
    eigenvalue qualifies (:302-313), so a channel without bound states reports n0_fin one higher than the one
    before.  Also the density-of-states factors rEki (:333-337; 1 where the reference leaves its initial value)."""
    import numpy as np
    E = np.asarray(E, dtype=np.float64)
    nl, nfun = E.shape
    n0_fin = -1; n1_fin = -1; nlim = 0; nbds = 0
    ntemp = 0                                   # the reference leaves it undefined until an eigenvalue <= Elim is seen
    out = StateLimits()
    out.n01 = np.zeros((nl, 3), dtype=np.int64); out.nbold = []; out.ntemp = []
    out.reki = np.ones((nl, nfun))
    for l in range(nl):
        En = E[l]
        if emax_fin == -1.0:
            emax_fin = float(En[nfun - 1]); elim = emax_fin
        else:
            elim = emax_fin + 0.25
            if kind_pi >= 8:
                elim = emax_fin
        i = 1; nbold = 0
        while True:
            e = En[i - 1]
            if e < 0.0:
                n0_fin = i; nbold += 1
            if e <= emax_fin:
                n1_fin = i
            if e <= elim:
                ntemp = i
            if e > emax_fin and e > elim:
                break
            i += 1
            if i > nfun:
                break
        nbds = max(nbds, nbold)
        n0_fin += 1; n1_fin += 1
        ne0 = n0_fin
        if kind_pi >= 5:
            n0_fin = 1
        nlim = max(nlim, ntemp)
        out.n01[l] = (n0_fin, n1_fin, ne0 - 1)
        out.nbold.append(nbold)
        ntemp = min(max(n1_fin + 40, nlim), nfun)
        out.ntemp.append(ntemp)
        # density of states: rEki(i) = sqrt(2/(E(i+1)-E(i-1))) inside, one-sided at nE0 and nfun
        if not 1 <= ne0 < nfun:
            raise ValueError("no continuum state in channel l=%d (the reference indexes En(%d) here)" % (l, ne0 + 1))
        for i in range(ne0 + 1, nfun):
            out.reki[l, i - 1] = math.sqrt(2.0 / (En[i] - En[i - 2]))
        out.reki[l, ne0 - 1] = math.sqrt(1.0 / (En[ne0] - En[ne0 - 1]))
        out.reki[l, nfun - 1] = math.sqrt(1.0 / (En[nfun - 1] - En[nfun - 2]))
    out.n1_max = min(max(int(out.n01[:, 1].max()) + 20, nlim), nfun)
    out.nbds = nbds; out.emax_fin = emax_fin
    if out.n1_max > out.ntemp[0]:
        # ctemp is allocated once, at l = 0, with ntemp(l=0) columns (:321-325): the reference would read past it
        raise ValueError("n1_max = %d exceeds the %d vectors the reference keeps (ctemp, matrices.f90:321)" % (out.n1_max, out.ntemp[0]))
    return out


def run(text, outdir=".", device=0, npts=10000):
    """`Bsp_Atom_omp.x < bsp_0.inp` on the MI355X: returns (E[lmax+1, nfun], c, stdout_text)."""
    inp = input_from_namelist(text)
    kind_pi = kind_pi_from_namelist(text)
    prob = capi.Problem(inp, device)
    out = ["PROGRAM TO CALCULATE ELECTRONIC STRUCTURE AND PI CROSS SECTIONS,".rjust(64), "  USING B-SPLINES", ""]
    out.append("Number of B-spline Functions / l: nfun =%5d" % prob.nfun)
    out.append("\nMax. Angular Momenta Included: l_max =%3d" % prob.lmax)
    E, info = prob.solve(0, prob.lmax + 1)
    c = None
    lim = None
    if kind_pi >= 3 and not any(info):
        lim = select_states(E, inp.emax_fin, kind_pi)
    with open(os.path.join(outdir, "Enl.dat"), "w") as f:
        f.write(" %d\n" % prob.nfun)                                   # WRITE(75,*) nfun
        for l in range(prob.lmax + 1):
            if info[l] != 0:
                out.append(" ERROR DIAGONALIZING THE MATRIX! %d" % info[l])
                out.append("\n l = %2d" % l)
                raise RuntimeError("\n".join(out[-2:]))
            out.append("\n l0 = %2d" % l)
            out.append(" HC = ESC eigenvalue solved\n")
            out.append("    n   Eigenvalues")
            out.append("    -   -----------")
            for i in range(prob.nfun):
                line = " %4d  %s" % (i + 1, fortran_g(E[l, i], 22, 15))      # FORMAT(T2,I4,T8,G22.15)
                if i < 20:
                    out.append(" %4d  %s" % (i + 1 + l, fortran_g(E[l, i], 22, 15)))
                f.write(line + "\n")
            if l == inp.l_ini:
                out.append("\nWriting down Initial State WF\n")
                c = prob.eigvec(l, inp.n0_ini)
                r, u = prob.write_wf(c, npts)                          # raises where the reference STOPs
                with open(os.path.join(outdir, "wf_n0.dat"), "w") as g:
                    for ri, ui in zip(r, u):
                        g.write(fortran_g(ri, 20, 10) + fortran_g(ui, 20, 10) + "\n")   # '(2G20.10)'
            if lim is not None:
                out.append("\nNUMBER OF BOUND STATES:%3d" % lim.nbold[l])                 # '(/,A23,I3)'
                out.append("LIMITS FOR l =%3d STATE%5d%5d" % (l, lim.n01[l, 0] + l, lim.n01[l, 1] + l))   # '(A14,I3,A6,2I5)'
    if lim is not None:
        out.append("n1_max =%5d" % lim.n1_max)                                          # '(A8,I5)'
        write_eigenvec_all(os.path.join(outdir, "Eigenvec_All.dat"), prob, prob.lmax, lim.n1_max)
    prob.close()
    os.makedirs(os.path.join(outdir, "CSs"), exist_ok=True)            # Bsp_Atom.f90:59-60 (`mkdir CSs` at start-up)
    if kind_pi in (1, 2):                                              # Bsp_Atom.f90:77-92: TRANS_AMP, CROSS_SECTIONS
        r = cross_sections(text, outdir=outdir, device=device)
        out.append("\n" + r["stdout"])
    if kind_pi in (0, 1, 2):
        out.append("\nProgram Finished!")          # KIND_PI >= 3 continues into the Gaussian / LG-beam branch in the reference
    return E, c, "\n".join(out)


def write_eigenvec_all(path, prob, lmax, n1_max):
    """`Eigenvec_All.dat` as SOLVE_SYSTEM writes it for KIND_PI >= 3 (matrices.f90:366-378): a list-directed
    header `nfun n1_max lmax`, then per channel a list-directed `l` and n1_max records FORMAT(I5,5000G20.10)
    `ni, c(1:nfun)`; the reader is READ_EIGENVEC (ReadInputs.f90:792-830).  n1_max comes from `select_states`;
    channels 0..lmax must have been solved.  List-directed integers are written the way flang does (one blank,
    no padding), which is what the golden files hold; gfortran/ifort pad them, READ(*,*) accepts either."""
    nfun = prob.nfun
    with open(path, "w") as f:
        f.write(" %d %d %d\n" % (nfun, n1_max, lmax))                  # WRITE(80,*) nfun, n1_max, lmax
        for l in range(lmax + 1):
            f.write(" %d\n" % l)
            Z = prob.eigvecs(l, 1, n1_max)
            for ni in range(n1_max):
                f.write("%5d" % (ni + 1) + "".join(fortran_g(v, 20, 10) for v in Z[ni]) + "\n")


def read_eigenvec_all(path):
    """Reads the file back the way READ_EIGENVEC does: returns (nfun, n1_max, lmax, c[l][ni][i])."""
    import numpy as np
    with open(path) as f:
        nfun, n1, lmax = (int(t) for t in f.readline().split())
        c = np.zeros((lmax + 1, n1, nfun))
        for l in range(lmax + 1):
            assert int(f.readline().split()[0]) == l
            for ni in range(n1):
                line = f.readline()
                assert int(line[:5]) == ni + 1
                c[l, ni] = [float(line[5 + 20 * i: 25 + 20 * i]) for i in range(nfun)]
    return nfun, n1, lmax, c



def read_enl(path, lmax, emax_fin=-1.0):
    """Reads `Enl.dat` back the way the reference's consumers do (READ_FR, ReadInputs.f90:292-316): list-directed
    `nfun`, then (lmax+1)*nfun records `i, En`; returns (nfun, Enl[l][i], n01[l]) with n01 = (1, n1_fin, n0_fin) as
    that reader rebuilds it (last index with En <= Emax_fin, last index with En < 0, both carried over between
    channels and starting undefined in the reference: 0 here)."""
    import numpy as np
    with open(path) as f:
        nfun = int(f.readline().split()[0])
        E = np.zeros((lmax + 1, nfun)); n01 = np.zeros((lmax + 1, 3), dtype=np.int64)
        n0_fin = 0; n1_fin = 0
        for l in range(lmax + 1):
            for ni in range(1, nfun + 1):
                t = f.readline().replace(",", " ").split()
                if len(t) < 2:
                    raise ValueError("Error Reading Energies")
                e = float(t[1].replace("D", "E").replace("d", "e"))
                E[l, ni - 1] = e
                if e < 0.0:
                    n0_fin = ni
                if e <= emax_fin:
                    n1_fin = ni
            n01[l] = (1, n1_fin, n0_fin)
    return nfun, E, n01


# ---- KIND_PI = 1, 2: transition amplitudes of the one-photon (plane wave) branches ---------------------------------
def three_j(j1, j2, j3, m1, m2, m3):
    """Wigner 3j symbol, integer arguments: Racah's sum evaluated through log-factorials with the smallest exponent
    taken out (what THREE_J does, Funs_WignerSymbols.for:1-62, so that the two agree to rounding)."""
    if m1 + m2 + m3 != 0:
        return 0.0
    lf = [0.0]
    for i in range(1, j1 + j2 + j3 + 2):
        lf.append(lf[-1] + math.log(float(i)))          # lf[i] = log(i!)
    zmin = max(0, j2 - j3 - m1, j1 + m2 - j3)
    zmax = min(j1 + j2 - j3, j1 - m1, j2 + m2)
    if zmax < zmin:
        return 0.0
    delta = 0.5 * (lf[j1 + j2 - j3] + lf[j3 + j1 - j2] + lf[j3 + j2 - j1] - lf[j1 + j2 + j3 + 1]
                   + lf[j1 + m1] + lf[j1 - m1] + lf[j2 + m2] + lf[j2 - m2] + lf[j3 + m3] + lf[j3 - m3])
    ex = [lf[z] + lf[j1 + j2 - j3 - z] + lf[j1 - m1 - z] + lf[j2 + m2 - z] + lf[j3 - j2 + m1 + z] + lf[j3 - j1 - m2 + z]
          for z in range(zmin, zmax + 1)]
    g = min(min(ex), 250.0)
    acc = 0.0
    for z, e in zip(range(zmin, zmax + 1), ex):
        acc += (-1.0) ** z * math.exp(-(e - g))
    return (-1.0) ** (j1 - j2 - m3) * math.exp(delta - g) * acc


def final_channels(kind_pi, l0, m0):
    """(l, m) of the final states SEL_LM selects for the dipolar cases (grid.f90:128-143): l0-1 (if it exists and can
    carry m0) and l0+1, m unchanged; TRANS_AMP uses the LAST of them."""
    out = []
    for lf in (l0 - 1, l0 + 1):
        if lf >= 0 and lf >= m0:
            out.append((lf, m0))
    return out


def final_state_limits(E_fin, emax_fin):
    """n0_fin, n1_fin (1-based) of SOLVE_SYSTEM for KIND_PI = 1, 2 at l = l_fin (matrices.f90:272-283) and the
    Emax_fin it then uses (-1 means the largest eigenvalue)."""
    n = len(E_fin)
    if emax_fin == -1.0:
        emax_fin = float(E_fin[n - 1])
    n0 = -1; n1 = -1
    for i in range(1, n + 1):
        if E_fin[i - 1] < 0.0:
            n0 = i
        if E_fin[i - 1] <= emax_fin:
            n1 = i
    return min(n0 + 1, n - 1), n1, emax_fin


def trans_amp(text, device=0):
    """`Bsp_Atom.x < input` with KIND_PI = 1 (length gauge) or 2 (velocity gauge) up to the end of TRANS_AMP
    (Bsp_Atom.f90:72-80; PhotoIon.f90:1-107): spectra of l = 0 .. lmax on the MI355X, the final-state window, and
    T_fi(n) = An c0 <c_fin(n)| c1 rij1 + c2 rij2 |c_ini> with the dipole matrices of the same assembly pass.
    Returns dict(E, l_fin, m_fin, n0_fin, n1_fin, T_fi, stdout).  The eigenvectors carry this library's sign
    convention (LAPACK's is arbitrary), so T_fi(n) agrees with the reference up to the sign of each state.
    CROSS_SECTIONS is not reproduced: the reference computes there with Enl(n0,l0), which it allocates for
    KIND_PI >= 3 only (PhotoIon.f90:302)."""
    import numpy as np
    nl = read_namelists(text)
    kind_pi = int(nl["vars_field"].get("kind_pi", 0))
    if kind_pi not in (1, 2):
        raise ValueError("trans_amp: KIND_PI must be 1 or 2")
    mph = int(nl["vars_field"].get("mph", 0))
    kw = {}
    kw.update(nl["vars_bsp"]); kw.update(nl["vars_tise"])
    inp = capi.make_input(**kw)
    l0, m0, n0 = inp.l_ini, inp.m_ini, inp.n0_ini
    lf, mf = final_channels(kind_pi, l0, m0)[-1]
    prob = capi.Problem(inp, device)
    if lf > prob.lmax:
        prob.close()
        raise ValueError("l_fin = l_ini + 1 = %d is beyond lmax = %d of the input (the reference reads an unallocated "
                         "ci_fin in that case)" % (lf, prob.lmax))
    E, info = prob.solve(0, prob.lmax + 1)
    if any(info):
        prob.close()
        raise RuntimeError("ERROR DIAGONALIZING THE MATRIX! %s" % list(info))
    n0f, n1f, _ = final_state_limits(E[lf], inp.emax_fin)
    if n1f + 1 > prob.nfun:
        prob.close()
        raise ValueError("n1_fin = nfun: the density-of-states factor needs E_fin(n1_fin + 1)")
    t3a = three_j(lf, 1, l0, -mf, mph, m0)
    if kind_pi == 1:
        t3b = three_j(lf, 1, l0, 0, 0, 0)
        c1 = (-1.0) ** (lf + l0 + mf) * math.sqrt(float((2 * lf + 1) * (2 * l0 + 1))) * t3a * t3b
        c0 = 1.0
        a = [c1, 0.0, 0.0]                                         # A = c1 * int B r B
    else:
        c0 = math.sqrt(float(l0 + 1)) * t3a
        c1, c2 = (float(l0 + 1), -1.0) if lf == l0 + 1 else ((float(l0), 1.0) if lf == l0 - 1 else (0.0, 0.0))
        a = [0.0, c1, c2]                                          # A = c1 * int B B / r + c2 * int B B'
    D = prob.dipole_elements(l0, n0, lf, n0f, n1f - n0f + 1, a)
    Ef = E[lf]
    T = np.array([math.sqrt(2.0 / (Ef[ni] - Ef[ni - 2])) * c0 * D[ni - n0f] for ni in range(n0f, n1f + 1)])
    out = ["LIMITS FOR FINAL STATE (l=%2d) : %4d%4d" % (lf, n0f, n1f),                 # '(/,A26,I2,A3,X,2I4)'
           "Calculating Transition Amplitudes",
           "Initial State:%3d%3d%3d" % (n0 + l0, l0, m0)]                              # '(A14,3I3)'
    prob.close()
    return dict(E=E, l_fin=lf, m_fin=mf, n0_fin=n0f, n1_fin=n1f, T_fi=T, stdout="\n".join(out))


# a.u. -> Mb and the speed of light of the reference (Modules.f90:12)
C_AU = 137.03599913815
A_AU = 5.29177249e-9


def cross_sections(text, outdir=".", device=0):
    """KIND_PI = 1, 2 to the end of CROSS_SECTIONS (PhotoIon.f90:274-468): `CSs/CrossSection_Len.dat` (length gauge) or
    `CSs/CrossSection_Vel.dat` (velocity gauge), one record FORMAT(2G20.10E3) `E_fin(nf), sigma(nf)` per final state
    nf = n0_fin .. n1_fin, sigma = M_au c0 c1 d1 T_fi(nf)^2 with M_au = a_au^2 1e18 (Mb), c0 = 4 pi^2 / c_au,
    c1 = 1/(2 l0 + 1), d1 = E_fin(nf) - E_ini(n0) (length) or its reciprocal (velocity) (:316-322, :387-394, :403).
    Two deviations from the reference AS WRITTEN, both where it reads variables SOLVE_SYSTEM sets for KIND_PI >= 3
    only: the loop `DO nf = n0_fin, n1_max` (:385) runs to n1_fin (the records are written for nf <= n1_fin anyway,
    :408), and the printed `E0=` (:302-303, Enl(n0,l0): unallocated here) is E_ini(n0).  The fixtures
    tests/golden/cs_*.npz come from the reference's own routine called with those two values set by the dump driver.
    T_fi is squared, so the arbitrary eigenvector signs drop out."""
    r = trans_amp(text, device=device)
    nl = read_namelists(text)
    kind_pi = int(nl["vars_field"]["kind_pi"])
    n0 = int(nl["vars_tise"].get("n0_ini", 1)); l0 = int(nl["vars_tise"].get("l_ini", 0))
    E_ini = r["E"][l0]; E_fin = r["E"][r["l_fin"]]
    m_au = (A_AU ** 2) * 1.0e18
    c0 = 4.0 * (math.pi ** 2) / C_AU
    c1 = 1.0 / float(2 * l0 + 1)
    rows = []
    for nf in range(r["n0_fin"], r["n1_fin"] + 1):
        d1 = E_fin[nf - 1] - E_ini[n0 - 1]
        if kind_pi == 2:
            d1 = 1.0 / d1
        d2 = r["T_fi"][nf - r["n0_fin"]] ** 2
        rows.append((E_fin[nf - 1], m_au * c0 * c1 * d1 * d2))
    os.makedirs(os.path.join(outdir, "CSs"), exist_ok=True)
    name = "CrossSection_Len.dat" if kind_pi == 1 else "CrossSection_Vel.dat"
    with open(os.path.join(outdir, "CSs", name), "w") as f:
        for ef, cs in rows:
            f.write(fortran_g(ef, 20, 10, 3) + fortran_g(cs, 20, 10, 3) + "\n")
    r["rows"] = rows
    r["file"] = os.path.join("CSs", name)
    r["stdout"] = r["stdout"] + "\n\nCalculating Cross Sections\n E0= %s" % repr(float(E_ini[n0 - 1]))
    return r


# ---- CSs/MatElem_All.dat: the coupling file the sibling TDSE tools read ------------------------------
def write_matelem_all(path, n1_max, zT):
    """`CSs/MatElem_All.dat` as TRANS_AMP writes it (PhotoIon.f90:255-266): list-directed header `n1_max nbra nket`, then
    for ibra = 1..nbra, jket = ibra..nket one record FORMAT(2I8,X,20G20.10) `ibra, jket, (Re, Im of zT(ibra,jket,i),
    i = 1..ncomp)`.  zT: complex array (nbra, nket, ncomp), upper triangle used.  The reader is READ_COUP
    (ReadInputs.f90:324-366)."""
    nbra, nket, ncomp = zT.shape
    with open(path, "w") as f:
        f.write(" %d %d %d\n" % (n1_max, nbra, nket))
        for ib in range(nbra):
            for jk in range(ib, nket):
                rec = "%8d%8d " % (ib + 1, jk + 1)
                for i in range(ncomp):
                    z = complex(zT[ib, jk, i])
                    rec += fortran_g(z.real, 20, 10) + fortran_g(z.imag, 20, 10)
                f.write(rec + "\n")


def read_matelem_all(path, nfields=1):
    """Reads the file back the way READ_COUP does (ReadInputs.f90:324-366): list-directed `n1_max nbra nket`, then
    list-directed records `ibra jket f(1:2 nfields)`; returns (n1_max, zHint[nbra][nket][nfields]), upper triangle."""
    import numpy as np
    with open(path) as f:
        n1_max, nbra, nket = (int(t) for t in f.readline().split())
        z = np.zeros((nbra, nket, nfields), dtype=np.complex128)
        for ni in range(nbra):
            for nj in range(ni, nket):
                t = f.readline().replace(",", " ").split()
                if len(t) < 2 + 2 * nfields:
                    raise ValueError("Error Reading Couplings File")
                ib, jk = int(t[0]), int(t[1])
                v = [float(x.replace("D", "E")) for x in t[2: 2 + 2 * nfields]]
                for i in range(nfields):
                    z[ib - 1, jk - 1, i] = complex(v[2 * i], v[2 * i + 1])
    return n1_max, z


def dipole_matelem(prob, channels, n1_max, kind_pi=1, mph=0):
    """Coupling matrix of the one-photon dipole operator between the states (l, n = 1..n1_max) of `channels` (a list of
    (l, m)), laid out as TRANS_AMP lays out zT_fi for the beam cases: row / column index = il * n1_max + n (il = position
    in `channels`), one component.  <n l m| c1 r |n' l' m'> in the length gauge (KIND_PI = 1) or the velocity-gauge
    operator (KIND_PI = 2) with the angular factors of PhotoIon.f90:66-83; zero unless l' = l +- 1.  The reference
    writes MatElem_All.dat only for its Gaussian / LG-beam branches (angular integrals outside SURVEY 8); this fills
    the same file with the plane-wave couplings so that READ_COUP consumers can run on the GPU solver's output."""
    import numpy as np
    nlm = len(channels)
    z = np.zeros((nlm * n1_max, nlm * n1_max, 1), dtype=np.complex128)
    for a_, (li, mi) in enumerate(channels):
        for b_, (lj, mj) in enumerate(channels):
            if abs(li - lj) != 1 or b_ < a_:
                continue
            # <bra = (li, mi)| A |ket = (lj, mj)>: the reference's c0, c1, c2 with l0 = lj (initial), lf = li (final)
            l0, m0, lf, mf = lj, mj, li, mi
            t3a = three_j(lf, 1, l0, -mf, mph, m0)
            if kind_pi == 1:
                t3b = three_j(lf, 1, l0, 0, 0, 0)
                c1 = (-1.0) ** (lf + l0 + mf) * math.sqrt(float((2 * lf + 1) * (2 * l0 + 1))) * t3a * t3b
                c0 = 1.0; coef = [c1, 0.0, 0.0]
            else:
                c0 = math.sqrt(float(l0 + 1)) * t3a
                c1, c2 = (float(l0 + 1), -1.0) if lf == l0 + 1 else (float(l0), 1.0)
                coef = [0.0, c1, c2]
            for nj in range(1, n1_max + 1):
                D = prob.dipole_elements(l0, nj, lf, 1, n1_max, coef)
                z[a_ * n1_max: (a_ + 1) * n1_max, b_ * n1_max + nj - 1, 0] = c0 * D
    return z


# ---- KIND_PI = 0 on several GPUs: one process per GPU, channels sharded, spectra gathered ---------------
def write_structure_outputs(nfun, lmax, E, l_ini, wf, outdir):
    """Enl.dat, wf_n0.dat and the stdout text of a KIND_PI = 0 run from the spectra E[l][i] and the tabulated initial
    state wf = (r, u) (matrices.f90:239-265,388-391; Bsp_Atom.f90:118-146) -- the formatting half of `run`."""
    out = ["PROGRAM TO CALCULATE ELECTRONIC STRUCTURE AND PI CROSS SECTIONS,".rjust(64), "  USING B-SPLINES", ""]
    out.append("Number of B-spline Functions / l: nfun =%5d" % nfun)
    out.append("\nMax. Angular Momenta Included: l_max =%3d" % lmax)
    with open(os.path.join(outdir, "Enl.dat"), "w") as f:
        f.write(" %d\n" % nfun)
        for l in range(lmax + 1):
            out.append("\n l0 = %2d" % l)
            out.append(" HC = ESC eigenvalue solved\n")
            out.append("    n   Eigenvalues")
            out.append("    -   -----------")
            for i in range(nfun):
                txt = fortran_g(E[l][i], 22, 15)
                if i < 20:
                    out.append(" %4d  %s" % (i + 1 + l, txt))
                f.write(" %4d  %s\n" % (i + 1, txt))
            if l == l_ini:
                out.append("\nWriting down Initial State WF\n")
                with open(os.path.join(outdir, "wf_n0.dat"), "w") as g:
                    for ri, ui in zip(*wf):
                        g.write(fortran_g(ri, 20, 10) + fortran_g(ui, 20, 10) + "\n")
    os.makedirs(os.path.join(outdir, "CSs"), exist_ok=True)
    out.append("\nProgram Finished!")
    return "\n".join(out)


def run_sharded(text, outdir=".", npts=10000, solver=None):
    """`Bsp_Atom_omp.x < bsp_0.inp` (KIND_PI = 0) on the GPUs of one node: started once per GPU by
    `python -m torch.distributed.run --nproc-per-node N -m bspatom_amd.host < bsp_0.inp`.  Rank r solves the block of
    l-channels `parallel.channel_range` gives it (the channels are independent, matrices.f90:242-248), the spectra are
    all-gathered (RCCL with backend nccl, gloo without a GPU), the rank that owns l_ini computes the consumed eigenvector
    and its WRITE_WF table and broadcasts them, rank 0 writes the files.  Output identical to `run`.
    solver(l0, nl) -> (E[nl][nfun], wf-or-None) replaces the GPU solve in the CPU tests."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from . import parallel
    if kind_pi_from_namelist(text) != 0:
        raise ValueError("run_sharded covers KIND_PI = 0 (the hot path); the other branches run on one GPU")
    inp = input_from_namelist(text)
    sizes = capi.host_setup(inp, arrays=False)
    nfun, lmax = sizes.nfun, sizes.lmax
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    counts = [parallel.channel_range(r, world, lmax)[1] for r in range(world)]
    l0, nl = parallel.channel_range(rank, world, lmax)
    owner = next(r for r in range(world) if sum(counts[:r]) <= inp.l_ini < sum(counts[:r + 1]))
    use_gpu = solver is None
    dev = torch.device("cuda", torch.cuda.current_device()) if use_gpu else torch.device("cpu")
    wf = None
    if use_gpu:
        prob = capi.Problem(inp, dev.index)
        E_loc = torch.zeros(max(nl, 1) * nfun, dtype=torch.float64, device=dev)
        if nl > 0:
            info = prob.solve_dev(l0, nl, E_loc.data_ptr())
            if info.any():
                raise RuntimeError(" ERROR DIAGONALIZING THE MATRIX! %s (l = %d ..)" % (list(info), l0))
        if rank == owner:
            c = prob.eigvec(inp.l_ini, inp.n0_ini)
            wf = prob.write_wf(c, npts)
        prob.close()
    else:
        E_np, wf = solver(l0, nl)
        E_loc = torch.zeros(max(nl, 1) * nfun, dtype=torch.float64)
        E_loc[: nl * nfun] = torch.from_numpy(np.ascontiguousarray(E_np, dtype=np.float64).reshape(-1))
    E_all = parallel.gather_spectra(E_loc[: nl * nfun], nfun, counts)
    wft = torch.zeros(2 * (npts + 1), dtype=torch.float64, device=dev)
    if rank == owner:
        wft[: npts + 1] = torch.from_numpy(np.asarray(wf[0])).to(dev)
        wft[npts + 1:] = torch.from_numpy(np.asarray(wf[1])).to(dev)
    if world > 1:
        dist.broadcast(wft, src=owner)
    if rank != 0:
        return None
    E = E_all.cpu().numpy()
    w = wft.cpu().numpy()
    return E, write_structure_outputs(nfun, lmax, E, inp.l_ini, (w[: npts + 1], w[npts + 1:]), outdir)


def _main():
    """torchrun entry: namelist on stdin of EVERY rank (torch.distributed.run forwards stdin to rank 0 only, so the text
    can also come from the file named by BSPATOM_INPUT)."""
    import sys
    import torch
    import torch.distributed as dist
    path = os.environ.get("BSPATOM_INPUT")
    text = open(path).read() if path else sys.stdin.read()
    launched = "RANK" in os.environ and "MASTER_ADDR" in os.environ
    # RCCL writes its banner to stdout when the communicator is created: keep the program's stdout clean (it is compared
    # with the reference's), everything else goes to stderr
    real_stdout = os.dup(1)
    sys.stdout.flush()
    os.dup2(2, 1)
    if launched:
        local = int(os.environ.get("LOCAL_RANK", "0"))
        if torch.cuda.is_available():
            torch.cuda.set_device(local)
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
        else:
            raise SystemExit("bspatom_amd.host needs an MI355X per rank: libbspatom has no CPU path")
    r = run_sharded(text, outdir=os.environ.get("BSPATOM_OUTDIR", "."))
    if r is not None:
        sys.stdout.flush()
        os.write(real_stdout, (r[1] + "\n").encode())
    if launched:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    _main()
