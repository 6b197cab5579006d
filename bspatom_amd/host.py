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


def fortran_g(v, w, d):
    """Fortran Gw.d edit descriptor (F2008 10.7.5.2.2) for a real value, as gfortran/flang print it."""
    if v == 0.0:
        body = "%.*f" % (d - 1, 0.0)
        return (body + "    ").rjust(w)
    a = abs(v)
    e = math.floor(math.log10(a)) + 1
    # rounding may push the value into the next decade
    if float("%.*e" % (d - 1, a)) >= 10.0 ** e:
        e += 1
    if 0 <= e <= d:
        body = "%.*f" % (d - e, v)
        if body.startswith("0."):
            pass
        return (body + "    ").rjust(w)
    m = "%.*E" % (d - 1, v)            # d.ddddE+xx -> 0.ddddd E+(xx+1)
    mant, ex = m.split("E")
    sign = "-" if mant.startswith("-") else ""
    digits = mant.replace("-", "").replace(".", "")
    ex = int(ex) + 1
    return ("%s0.%sE%+03d" % (sign, digits, ex)).rjust(w)


def kind_pi_from_namelist(text):
    return int(read_namelists(text)["vars_field"].get("kind_pi", 0))


def input_from_namelist(text):
    nl = read_namelists(text)
    if nl["vars_field"].get("kind_pi", 0) in (1, 2):
        raise ValueError("KIND_PI = 1, 2 (one-photon cross sections) continue into TRANS_AMP, which is not on the "
                         "MI355X hot path; use KIND_PI = 0 or >= 3 (SOLVE_SYSTEM only)")
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
    if kind_pi == 0:
        out.append("\nProgram Finished!")          # KIND_PI >= 3 continues into the photo-ionisation branch in the reference
    prob.close()
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
