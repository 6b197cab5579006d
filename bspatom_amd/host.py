"""Python host of libbspatom mirroring the reference driver for KIND_PI = 0
(PROGRAM BSP_ATOM_PI, src/Bsp_Atom.f90:45-95; SOLVE_SYSTEM output, src/matrices.f90:239-267):
namelist text -> spectra, the (l_ini, n0_ini) eigenvector, Enl.dat / wf_n0.dat / stdout text in the
reference's formats.  All arithmetic happens in libbspatom on the GPU; this module only parses,
dispatches and formats."""
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


def input_from_namelist(text):
    nl = read_namelists(text)
    if nl["vars_field"].get("kind_pi", 0) != 0:
        raise ValueError("only KIND_PI = 0 (electronic structure) is on the MI355X hot path")
    kw = {}
    kw.update(nl["vars_bsp"]); kw.update(nl["vars_tise"])
    return capi.make_input(**kw)


def run(text, outdir=".", device=0, npts=10000):
    """`Bsp_Atom_omp.x < bsp_0.inp` on the MI355X: returns (E[lmax+1, nfun], c, stdout_text)."""
    inp = input_from_namelist(text)
    prob = capi.Problem(inp, device)
    out = ["PROGRAM TO CALCULATE ELECTRONIC STRUCTURE AND PI CROSS SECTIONS,".rjust(64), "  USING B-SPLINES", ""]
    out.append("Number of B-spline Functions / l: nfun =%5d" % prob.nfun)
    out.append("\nMax. Angular Momenta Included: l_max =%3d" % prob.lmax)
    E, info = prob.solve(0, prob.lmax + 1)
    c = None
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
    out.append("\nProgram Finished!")
    prob.close()
    return E, c, "\n".join(out)


def write_eigenvec_all(path, prob, lmax, n1_max):
    """`Eigenvec_All.dat` as SOLVE_SYSTEM writes it for KIND_PI >= 3 (matrices.f90:366-378): a list-directed
    header `nfun n1_max lmax`, then per channel a list-directed `l` and n1_max records FORMAT(I5,5000G20.10)
    `ni, c(1:nfun)`; the reader is READ_EIGENVEC (ReadInputs.f90:792-830).  The caller chooses n1_max (the
    reference derives it from Emax_fin in its photo-ionisation branch, which is outside the hot path) and must
    have solved channels 0..lmax."""
    nfun = prob.nfun
    with open(path, "w") as f:
        f.write(" %11d %11d %11d\n" % (nfun, n1_max, lmax))            # WRITE(80,*) of three default integers
        for l in range(lmax + 1):
            f.write(" %11d\n" % l)
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

