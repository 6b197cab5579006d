#!/usr/bin/env python3
"""make_golden.py -- generate tests/golden/*.npz from the COMPILED REFERENCE (oracle/_ref).

TEST INFRASTRUCTURE.  Run in the build container only (needs /root/reference, through
oracle/ref/build_ref.sh).  For every namelist in tests/golden/inputs/ it runs

  oracle/_ref/ref_dump.x      -> ref_dump.bin : module state after GRID + MATRIX_SVT
                                 (rt, xg, wg, Aind, Sij, Tij, Vij, Uij), then SOLVE_SYSTEM
  (same run)                  -> Enl.dat, wf_n0.dat, stdout

and stores DATA ONLY (inputs + expected outputs) as compressed .npz:
  sizes, rt, xg, wg, aind, upper bands of S/T/V/U_l (matrix cases), all eigenvalues per l,
  every 50th row of wf_n0.dat, stdout eigenvalue lines.
Manifest (compiler, LAPACK) is written to tests/golden/MANIFEST.json.

usage: python tests/golden/make_golden.py [case ...]      (default: all small cases)
       python tests/golden/make_golden.py --big            (adds the n=2048/4096 spectra)
       python tests/golden/make_golden.py --dipole         (KIND_PI = 1, 2 dipole matrices rij)
       python tests/golden/make_golden.py --pi3            (KIND_PI >= 3: state limits, rEki, Eigenvec_All.dat)
       python tests/golden/make_golden.py --amp            (KIND_PI = 1, 2: transition amplitudes T_fi of TRANS_AMP)
       python tests/golden/make_golden.py --vectors        (eigenvectors of DSYGV('V') at n = 2048: vec_c2_2048.npz)
"""
import json, os, subprocess, sys, tempfile, time
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REFX = os.path.join(ROOT, "oracle", "_ref", "ref_dump.x")

FIELD = "&VARS_FIELD KIND_PI=0 &end\n"

def nml(bsp, tise):
    return "&VARS_BSP %s &end\n&VARS_TISE %s &end\n%s" % (bsp, tise, FIELD)

# name -> (namelist text, store_matrices)
CASES = {
    # the reference's only fixture, verbatim values of exec/bsp_0.inp (nfun 100 -> 124)
    "bsp0": (None, True),
    # C1 plumbing cases: exponential and linear grids, k=7, nfun=64
    "c1_exp": (nml("KIND_GRID=1 ra=0.0D0 rb=200.0D0 k=7 nfun=64", "n0_ini=1 l_ini=0 l_fin=0 Zatom=1.0D0"), True),
    "c1_lin": (nml("KIND_GRID=0 ra=0.0D0 rb=50.0D0 k=7 nfun=64", "n0_ini=2 l_ini=1 l_fin=3 Zatom=1.0D0"), True),
    # Rogers screened-Coulomb (KIND_POT=1), Zatom=20
    "rogers": (nml("KIND_GRID=1 ra=0.0D0 rb=100.0D0 k=7 nfun=96", "n0_ini=1 l_ini=0 l_fin=1 Zatom=20.0D0 KIND_POT=1"), True),
    # Simons-Fues (KIND_POT=2): l-dependent Bl/r^2 term, lmax > 3 exercises Bl(l>3)=0
    "simfues": (nml("KIND_GRID=0 ra=0.0D0 rb=60.0D0 k=8 nfun=80", "n0_ini=1 l_ini=0 l_fin=4 Zatom=1.0D0 KIND_POT=2"), True),
    # KIND_BC1=1 / KIND_BC2=1: first and last B-spline kept (knot multiplicity k)
    "bc1": (nml("KIND_GRID=0 ra=0.0D0 rb=30.0D0 k=5 nfun=40 KIND_BC1=1 KIND_BC2=1", "n0_ini=1 l_ini=0 l_fin=1 Zatom=1.0D0"), True),
    # explicit ka, ra /= 0
    "ka_ra": (nml("KIND_GRID=0 ra=0.5D0 rb=40.5D0 k=6 ka=8 nfun=48", "n0_ini=1 l_ini=0 l_fin=2 Zatom=2.0D0"), True),
    # reference edge case: with ra=0.5, rb=40 the last WRITE_WF point ra+10000*dr rounds above rb,
    # interv returns left=1 and BSPLVB STOPs ('FATAL ERROR - BSPLVB', bsplvb.f90:30-34) after l=l_ini
    "wf_fatal": (nml("KIND_GRID=0 ra=0.5D0 rb=40.0D0 k=6 ka=8 nfun=48", "n0_ini=1 l_ini=0 l_fin=2 Zatom=2.0D0"), True),
    # reduced-size analogues of C2-C5 (linear grids)
    "lin256": (nml("KIND_GRID=0 ra=0.0D0 rb=50.0D0 k=9 nfun=256", "n0_ini=1 l_ini=0 l_fin=3 Zatom=1.0D0"), True),
    "yuk256": (nml("KIND_GRID=0 ra=0.0D0 rb=25.0D0 k=11 nfun=256", "n0_ini=1 l_ini=0 l_fin=0 Zatom=20.0D0 KIND_POT=1"), True),
    "lin1024": (nml("KIND_GRID=0 ra=0.0D0 rb=200.0D0 k=9 nfun=1024", "n0_ini=2 l_ini=0 l_fin=1 Zatom=1.0D0"), False),
    # shapes at the edges of the kernels' tilings (round 2): nfun barely above k (one knot interval more than the order needs),
    # nfun = 65 and 128 (one row over a 64-block / exactly two blocks: one panel of the dense -> band stage), k = 4 (ka = 7, odd),
    # asymmetric boundary conditions (first B-spline kept, last one dropped)
    "tiny8": (nml("KIND_GRID=0 ra=0.0D0 rb=12.0D0 k=5 nfun=8", "n0_ini=1 l_ini=0 l_fin=1 Zatom=1.0D0"), True),
    "n65_k4": (nml("KIND_GRID=0 ra=0.0D0 rb=40.0D0 k=4 nfun=65", "n0_ini=1 l_ini=0 l_fin=1 Zatom=1.0D0"), True),
    "n128": (nml("KIND_GRID=0 ra=0.0D0 rb=60.0D0 k=8 nfun=128", "n0_ini=2 l_ini=1 l_fin=2 Zatom=2.0D0"), True),
    "bc10": (nml("KIND_GRID=0 ra=0.0D0 rb=30.0D0 k=6 nfun=50 KIND_BC1=1 KIND_BC2=0", "n0_ini=1 l_ini=0 l_fin=1 Zatom=1.0D0"), True),
}
BIG = {
    # C3 corners: channels l = 30, 31 of the 32-channel batch (lmax=31 would store 32 spectra; l_ini selects nothing new),
    # obtained by running the reference with l_fin=31 is 32 x 13 s; instead two single-channel-equivalent runs are not
    # possible in the reference (it always loops l = 0..lmax), so a reduced n=1024 batch l=0..31 is used for all 32 channels
    "c3_1024_l31": (nml("KIND_GRID=0 ra=0.0D0 rb=200.0D0 k=9 nfun=1024", "n0_ini=1 l_ini=0 l_fin=31 Zatom=1.0D0"), False),
    # C5-like: Rogers screened Coulomb, k=11 (ka=14), reduced n
    "c5_1024_k11": (nml("KIND_GRID=0 ra=0.0D0 rb=100.0D0 k=11 nfun=1024", "n0_ini=1 l_ini=0 l_fin=1 Zatom=20.0D0 KIND_POT=1"), False),
    # BASELINE configs[1] (C2) and one channel each of C4 (l=0 and l=3 via lmax), spectra only
    "c2_2048": (nml("KIND_GRID=0 ra=0.0D0 rb=400.0D0 k=9 nfun=2048", "n0_ini=1 l_ini=0 l_fin=0 Zatom=1.0D0"), False),
    "c4_4096": (nml("KIND_GRID=0 ra=0.0D0 rb=800.0D0 k=9 nfun=4096", "n0_ini=1 l_ini=0 l_fin=1 Zatom=1.0D0"), False),
    # BASELINE configs[4] (C5) AT ITS REAL SIZE: Rogers screened Coulomb (the reference's Yukawa-type potential), n=8192, k=11,
    # one channel; ~15 min of the reference's DSYGV.  n > 4096 takes the first panel kernel and rings of 8 in sb2st.
    "c5_8192": (nml("KIND_GRID=0 ra=0.0D0 rb=800.0D0 k=11 nfun=8192", "n0_ini=1 l_ini=0 l_fin=0 Zatom=20.0D0 KIND_POT=1"), False),
    # SURVEY 8(f).4 "at scale": the grid / boundary-condition / potential variants at nfun ~ 2048
    "exp2048": (nml("KIND_GRID=1 ra=0.0D0 rb=400.0D0 k=9 nfun=2048", "n0_ini=1 l_ini=0 l_fin=1 Zatom=1.0D0"), False),
    "explin2048": (nml("KIND_GRID=2 rmax=40.0D0 ra=0.0D0 rb=400.0D0 k=9 nfun=2048", "n0_ini=1 l_ini=0 l_fin=1 Zatom=1.0D0"), False),
    "bc1_2048": (nml("KIND_GRID=0 ra=0.0D0 rb=400.0D0 k=9 nfun=2048 KIND_BC1=1 KIND_BC2=1", "n0_ini=1 l_ini=0 l_fin=1 Zatom=1.0D0"), False),
    "sf2048": (nml("KIND_GRID=0 ra=0.0D0 rb=400.0D0 k=9 nfun=2048", "n0_ini=1 l_ini=0 l_fin=4 Zatom=1.0D0 KIND_POT=2"), False),
}

# dipole matrices rij that MATRIX_SVT keeps for KIND_PI = 1 (length) / 2 (velocity).  name -> (namelist, KIND_PI)
def nml_pi(bsp, tise, kind_pi):
    return "&VARS_BSP %s &end\n&VARS_TISE %s &end\n&VARS_FIELD KIND_PI=%d Eph=0.5D0 I0=1.0D14 &end\n" % (bsp, tise, kind_pi)

DIPOLE = {
    "dip_len_lin": (nml_pi("KIND_GRID=0 ra=0.0D0 rb=50.0D0 k=7 nfun=64", "n0_ini=1 l_ini=0 l_fin=1 Zatom=1.0D0", 1), 1),
    "dip_vel_lin": (nml_pi("KIND_GRID=0 ra=0.0D0 rb=50.0D0 k=7 nfun=64", "n0_ini=1 l_ini=0 l_fin=1 Zatom=1.0D0", 2), 2),
    "dip_len_exp": (nml_pi("KIND_GRID=1 ra=0.0D0 rb=200.0D0 k=9 nfun=96", "n0_ini=1 l_ini=0 l_fin=1 Zatom=1.0D0", 1), 1),
    "dip_vel_exp": (nml_pi("KIND_GRID=1 ra=0.0D0 rb=200.0D0 k=9 nfun=96", "n0_ini=1 l_ini=0 l_fin=1 Zatom=1.0D0", 2), 2),
}

def run_dipole(name, text, kind_pi):
    inp = os.path.join(HERE, "inputs", name + ".inp")
    with open(inp, "w") as f:
        f.write("! golden-fixture input '%s' (generated by make_golden.py)\n" % name)
        f.write(text)
    with tempfile.TemporaryDirectory(prefix="bspgold.") as tmp:
        with open(inp) as fin:
            p = subprocess.run([REFX], stdin=fin, cwd=tmp, capture_output=True, text=True)
        if p.returncode != 0:
            raise RuntimeError("reference failed on %s:\n%s\n%s" % (name, p.stdout[-2000:], p.stderr[-2000:]))
        raw = open(os.path.join(tmp, "ref_dump.bin"), "rb").read()
        hdr = np.frombuffer(raw[:32], dtype=np.int32)
        nfun, k, ka, nkp, nointv, nbc1, nbc2, lmax = [int(v) for v in hdr]
        rr = open(os.path.join(tmp, "ref_rij.bin"), "rb").read()
        h2 = np.frombuffer(rr[:8], dtype=np.int32)
        assert int(h2[0]) == nfun and int(h2[1]) == kind_pi
        R = np.frombuffer(rr[8:], dtype=np.float64).reshape(2, nfun, nfun)       # rij(:,:,m) column-major -> R[m][j][i]
        r1, r2 = R[0].T, R[1].T
        # MATRIX_SVT fills BOTH triangles (jket = 1..nfun, matrices.f90:69) and the two are not bit-symmetric
        # (((fbra*r)*fket)*dr vs ((fket*r)*fbra)*dr; int B_i B_j' is not symmetric at all): full band, 2k-1 diagonals
        def full_band(M):
            B = np.zeros((2 * k - 1, nfun))
            for d in range(-(k - 1), k):
                i = np.arange(max(0, -d), min(nfun, nfun - d))
                B[d + k - 1, i] = M[i, i + d]
            return B
        out = dict(sizes=np.array([nfun, k, ka, nkp, nointv, nbc1, nbc2, lmax]), kind_pi=np.array([kind_pi]),
                   r1f=full_band(r1), r2f=full_band(r2),
                   outside_band_max=np.array([max(np.max(np.abs(np.tril(r1, -k))), np.max(np.abs(np.tril(r2, -k))),
                                                  np.max(np.abs(np.triu(r1, k))), np.max(np.abs(np.triu(r2, k))))]),
                   namelist=np.array(open(inp).read()))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("%-12s nfun=%4d k=%2d KIND_PI=%d  |r1|max %.6g |r2|max %.6g outside band %.1e" % (
        name, nfun, k, kind_pi, np.max(np.abs(out["r1f"])), np.max(np.abs(out["r2f"])), out["outside_band_max"][0]), flush=True)

# KIND_PI >= 3 branch of SOLVE_SYSTEM (matrices.f90:290-378): state limits n01, n1_max, density-of-states
# factors rEki and the Eigenvec_All.dat file.  name -> namelist
def nml_pi3(bsp, tise, kind_pi):
    return "&VARS_BSP %s &end\n&VARS_TISE %s &end\n&VARS_FIELD KIND_PI=%d Eph=0.5D0 I0=1.0D14 &end\n" % (bsp, tise, kind_pi)

_B64 = "KIND_GRID=0 ra=0.0D0 rb=50.0D0 k=7 nfun=64"
PI3 = {
    "pi3_emax1": nml_pi3(_B64, "n0_ini=1 l_ini=0 l_fin=2 Emax_fin=1.0D0 Zatom=1.0D0", 3),
    "pi3_default": nml_pi3(_B64, "n0_ini=1 l_ini=0 l_fin=2 Zatom=1.0D0", 3),           # Emax_fin = -1 -> En(nfun) of l = 0
    "pi5_emax05": nml_pi3(_B64, "n0_ini=2 l_ini=1 l_fin=2 Emax_fin=0.5D0 Zatom=1.0D0", 5),   # KIND_PI >= 5: n0_fin = 1
    "pi8_emax1": nml_pi3(_B64, "n0_ini=1 l_ini=0 l_fin=2 Emax_fin=1.0D0 Zatom=1.0D0", 8),     # KIND_PI >= 8: Elim = Emax_fin
    # small box: no bound state for l >= 2, the limits carry over from the previous channel (matrices.f90:305-316)
    "pi3_nobound": nml_pi3("KIND_GRID=0 ra=0.0D0 rb=6.0D0 k=5 nfun=40", "n0_ini=1 l_ini=0 l_fin=3 Emax_fin=4.0D0 Zatom=1.0D0", 3),
}

def run_pi3(name, text):
    inp = os.path.join(HERE, "inputs", name + ".inp")
    with open(inp, "w") as f:
        f.write("! golden-fixture input '%s' (generated by make_golden.py)\n" % name)
        f.write(text)
    with tempfile.TemporaryDirectory(prefix="bspgold.") as tmp:
        with open(inp) as fin:
            p = subprocess.run([REFX], stdin=fin, cwd=tmp, capture_output=True, text=True)
        if p.returncode != 0:
            raise RuntimeError("reference failed on %s:\n%s\n%s" % (name, p.stdout[-2000:], p.stderr[-2000:]))
        raw = open(os.path.join(tmp, "ref_pi3.bin"), "rb").read()
        nfun, lmax, n1_max, kind_pi = [int(v) for v in np.frombuffer(raw[:16], dtype=np.int32)]
        off = 16
        n01 = np.frombuffer(raw[off: off + 4 * 3 * (lmax + 1)], dtype=np.int32).reshape(3, lmax + 1).T.copy(); off += 4 * 3 * (lmax + 1)
        emax = np.frombuffer(raw[off: off + 8], dtype=np.float64).copy(); off += 8
        nn = nfun * (lmax + 1)
        E = np.frombuffer(raw[off: off + 8 * nn], dtype=np.float64).reshape(lmax + 1, nfun).copy(); off += 8 * nn
        reki = np.frombuffer(raw[off: off + 8 * nn], dtype=np.float64).reshape(lmax + 1, nfun).copy(); off += 8 * nn
        assert off == len(raw)
        lines = open(os.path.join(tmp, "Eigenvec_All.dat")).read().split("\n")
        hdr = [int(t) for t in lines[0].split()]
        assert hdr == [nfun, n1_max, lmax]
        C = np.zeros((lmax + 1, n1_max, nfun))
        pos = 1
        for l in range(lmax + 1):
            assert int(lines[pos].split()[0]) == l; pos += 1
            for ni in range(n1_max):
                ln = lines[pos]; pos += 1
                assert int(ln[:5]) == ni + 1
                C[l, ni] = [float(ln[5 + 20 * i: 25 + 20 * i]) for i in range(nfun)]
        sel = [l for l in p.stdout.split("\n") if ("BOUND STATES" in l or "LIMITS FOR" in l or "n1_max" in l)]
        out = dict(sizes=np.array([nfun, lmax, n1_max, kind_pi]), n01=n01, emax_fin_out=emax, E=E, reki=reki, C=C,
                   eva_head=np.array("\n".join(lines[:2])), eva_row=np.array(lines[2]), limits=np.array("\n".join(sel)),
                   stdout=np.array(p.stdout), namelist=np.array(open(inp).read()))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("%-12s nfun=%3d lmax=%d KIND_PI=%d n1_max=%d n01=%s" % (name, nfun, lmax, kind_pi, n1_max, n01.tolist()), flush=True)

# transition amplitudes T_fi of TRANS_AMP for KIND_PI = 1, 2 (PhotoIon.f90:50-107).  l_fin must be l_ini + 1
# in the input (SEL_LM replaces it by that anyway, grid.f90:143, but lmax is derived from the input value before).
def nml_amp(bsp, tise, kind_pi):
    return "&VARS_BSP %s &end\n&VARS_TISE %s &end\n&VARS_FIELD KIND_PI=%d Eph=0.5D0 I0=1.0D14 &end\n" % (bsp, tise, kind_pi)

AMP = {
    "ta_len_s": nml_amp(_B64, "n0_ini=1 l_ini=0 l_fin=1 Emax_fin=1.0D0 Zatom=1.0D0", 1),       # 1s -> p, length gauge
    "ta_vel_s": nml_amp(_B64, "n0_ini=1 l_ini=0 l_fin=1 Emax_fin=1.0D0 Zatom=1.0D0", 2),       # velocity gauge
    "ta_len_p": nml_amp(_B64, "n0_ini=1 l_ini=1 l_fin=2 Emax_fin=0.6D0 Zatom=1.0D0", 1),       # 2p -> d
    "ta_vel_p": nml_amp("KIND_GRID=1 ra=0.0D0 rb=120.0D0 k=8 nfun=80", "n0_ini=2 l_ini=1 l_fin=2 Emax_fin=0.8D0 Zatom=2.0D0", 2),
}

def run_amp(name, text):
    inp = os.path.join(HERE, "inputs", name + ".inp")
    with open(inp, "w") as f:
        f.write("! golden-fixture input '%s' (generated by make_golden.py)\n" % name)
        f.write(text)
    with tempfile.TemporaryDirectory(prefix="bspgold.") as tmp:
        with open(inp) as fin:
            p = subprocess.run([REFX], stdin=fin, cwd=tmp, capture_output=True, text=True)
        if p.returncode != 0:
            raise RuntimeError("reference failed on %s:\n%s\n%s" % (name, p.stdout[-2000:], p.stderr[-2000:]))
        raw = open(os.path.join(tmp, "ref_tfi.bin"), "rb").read()
        h = [int(v) for v in np.frombuffer(raw[:40], dtype=np.int32)]
        nfun, kind_pi, n0_ini, l0, m0, lf, mf, mph, n0_fin, n1_fin = h
        off = 40
        def take(cnt):
            nonlocal off
            a = np.frombuffer(raw[off: off + 8 * cnt], dtype=np.float64).copy(); off += 8 * cnt
            return a
        emax = take(1); E_ini = take(nfun); E_fin = take(nfun); ci_ini = take(nfun)
        nf = n1_fin - n0_fin + 1
        ci_fin = take(nfun * nf).reshape(nf, nfun).T.copy()           # (nfun, nf)
        T = take(nf)
        assert off == len(raw)
        rr = open(os.path.join(tmp, "ref_rij.bin"), "rb").read()
        R = np.frombuffer(rr[8:], dtype=np.float64).reshape(2, nfun, nfun)
        sel = [l for l in p.stdout.split("\n") if ("LIMITS FOR FINAL" in l or "Initial State" in l or "Transition Amplitudes" in l)]
        # CSs/CrossSection_Len.dat / _Vel.dat written by the reference's CROSS_SECTIONS (see ref_dump_driver.f90)
        csname = "CrossSection_Len.dat" if kind_pi == 1 else "CrossSection_Vel.dat"
        cstext = open(os.path.join(tmp, "CSs", csname)).read()
        cs = np.array([[float(t) for t in ln.split()] for ln in cstext.split("\n") if ln.strip()])
        out = dict(head=np.array(h), emax_fin=emax, E_ini=E_ini, E_fin=E_fin, ci_ini=ci_ini, ci_fin=ci_fin, T_fi=T,
                   r1=R[0].T.copy(), r2=R[1].T.copy(), lines=np.array("\n".join(sel)), namelist=np.array(open(inp).read()),
                   cs_file=np.array(csname), cs_text=np.array(cstext), cs_rows=cs)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("%-10s KIND_PI=%d (n0,l0,m0)=(%d,%d,%d) -> (lf,mf)=(%d,%d) states %d..%d  max|T| %.6g" % (
        name, kind_pi, n0_ini, l0, m0, lf, mf, n0_fin, n1_fin, np.max(np.abs(T))), flush=True)

def upper_band(M, k):
    n = M.shape[0]
    B = np.zeros((k, n))
    for d in range(k):
        B[d, : n - d] = np.diagonal(M, d)
    # everything outside the band must be an exact zero in the reference's dense matrix
    mask = np.abs(np.subtract.outer(np.arange(n), np.arange(n))) >= k
    assert not np.any(M[mask]), "reference matrix has non-zeros outside the band"
    return B

def run_case(name, text, store_mats):
    inp = os.path.join(HERE, "inputs", name + ".inp")
    if text is not None:
        with open(inp, "w") as f:
            f.write("! golden-fixture input '%s' (generated by make_golden.py)\n" % name)
            f.write(text)
    with tempfile.TemporaryDirectory(prefix="bspgold.") as tmp:
        t0 = time.time()
        with open(inp) as fin:
            p = subprocess.run([REFX], stdin=fin, cwd=tmp, capture_output=True, text=True)
        wall = time.time() - t0
        fatal = ("FATAL ERROR" in p.stdout) or ("ERROR DIAGONALIZING" in p.stdout)
        if p.returncode != 0 and not fatal:
            raise RuntimeError("reference failed on %s:\n%s\n%s" % (name, p.stdout[-2000:], p.stderr[-2000:]))
        raw = open(os.path.join(tmp, "ref_dump.bin"), "rb").read()
        hdr = np.frombuffer(raw[:32], dtype=np.int32)
        nfun, k, ka, nkp, nointv, nbc1, nbc2, lmax = [int(v) for v in hdr]
        off = 32
        def take(cnt):
            nonlocal off
            a = np.frombuffer(raw[off: off + 8 * cnt], dtype=np.float64).copy()
            off += 8 * cnt
            return a
        rt = take(nkp); xg = take(ka); wg = take(ka)
        aind = take(2 * nfun)                      # column-major (nfun,2)
        nn = nfun * nfun
        out = dict(sizes=np.array([nfun, k, ka, nkp, nointv, nbc1, nbc2, lmax]), rt=rt, xg=xg, wg=wg, aind=aind)
        if store_mats:
            S = take(nn).reshape(nfun, nfun).T     # -> S[i,j]
            T = take(nn).reshape(nfun, nfun).T
            V = take(nn).reshape(nfun, nfun).T
            U = [take(nn).reshape(nfun, nfun).T for _ in range(lmax + 1)]
            out["Sb"] = upper_band(S, k); out["Tb"] = upper_band(T, k); out["Vb"] = upper_band(V, k)
            out["Ub"] = np.stack([upper_band(u, k) for u in U])
        # Enl.dat
        lines = open(os.path.join(tmp, "Enl.dat")).read().split("\n")
        assert int(lines[0]) == nfun
        vals = [float(l.split()[1]) for l in lines[1:] if l.strip()]
        out["fatal"] = np.array([int(fatal)])
        nl_done = len(vals) // nfun
        out["E"] = np.array(vals[: nl_done * nfun]).reshape(nl_done, nfun)
        if not fatal:
            assert nl_done == lmax + 1
            wf = np.loadtxt(os.path.join(tmp, "wf_n0.dat"))
            assert wf.shape == (10001, 2)
            out["wf_idx"] = np.arange(0, 10001, 50)
            out["wf_rows"] = wf[::50].copy()
            out["wf_norm2"] = np.array([np.sum(wf[:, 1] ** 2) * (wf[1, 0] - wf[0, 0])])
        tim = {}
        for l in p.stdout.split("\n"):
            if l.startswith("REF_TIME_"):
                tim[l.split()[0]] = float(l.split()[1])
        out["ref_time_assembly_s"] = np.array([tim.get("REF_TIME_MATRIX_SVT_S", np.nan)])
        out["ref_time_solve_s"] = np.array([tim.get("REF_TIME_SOLVE_SYSTEM_S", np.nan)])
        out["stdout"] = np.array(p.stdout)
        out["namelist"] = np.array(open(inp).read())
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("%-10s nfun=%5d k=%2d lmax=%2d  E1=%.15g  asm %.2fs solve %.2fs wall %.1fs" % (
        name, nfun, k, lmax, out["E"][0, 0], out["ref_time_assembly_s"][0], out["ref_time_solve_s"][0], wall), flush=True)

# Spectra-only fixtures from the reference PROGRAM itself (oracle/_ref/Bsp_Atom_ref.x, no dump driver: the dense dump of
# Uij(nfun,nfun,0:lmax) would be 17 GB at C4): BASELINE configs[2] and configs[3] at their real sizes, every channel.
SPECTRA = {
    # C3: Hydrogen l = 0..31 batched, N_bsp = 2048 (32 x ~13 s of DSYGV)
    "c3_2048_l31": nml("KIND_GRID=0 ra=0.0D0 rb=400.0D0 k=9 nfun=2048", "n0_ini=1 l_ini=0 l_fin=31 Zatom=1.0D0"),
    # C4: Hydrogen l = 0..127, N_bsp = 4096 -- the bench workload, all 128 channels (~3.5 h of DSYGV on 8 cores)
    "c4_4096_l127": nml("KIND_GRID=0 ra=0.0D0 rb=800.0D0 k=9 nfun=4096", "n0_ini=1 l_ini=0 l_fin=127 Zatom=1.0D0"),
}

def run_spectra(name, text):
    refprog = os.path.join(ROOT, "oracle", "_ref", "Bsp_Atom_ref.x")
    inp = os.path.join(HERE, "inputs", name + ".inp")
    with open(inp, "w") as f:
        f.write("! golden-fixture input '%s' (generated by make_golden.py)\n" % name)
        f.write(text)
    with tempfile.TemporaryDirectory(prefix="bspgold.") as tmp:
        t0 = time.time()
        with open(inp) as fin:
            p = subprocess.run([refprog], stdin=fin, cwd=tmp, capture_output=True, text=True)
        wall = time.time() - t0
        if p.returncode != 0 or "Program Finished!" not in p.stdout:
            raise RuntimeError("reference failed on %s:\n%s\n%s" % (name, p.stdout[-2000:], p.stderr[-2000:]))
        lines = open(os.path.join(tmp, "Enl.dat")).read().split("\n")
        nfun = int(lines[0])
        vals = np.array([float(l.split()[1]) for l in lines[1:] if l.strip()])
        E = vals.reshape(-1, nfun)
        wf = np.loadtxt(os.path.join(tmp, "wf_n0.dat"))
        out = dict(sizes=np.array([nfun, 0, 0, 0, 0, 0, 0, E.shape[0] - 1]), E=E, wf_idx=np.arange(0, 10001, 50), wf_rows=wf[::50].copy(),
                   ref_wall_s=np.array([wall]), namelist=np.array(open(inp).read()),
                   source=np.array("oracle/_ref/Bsp_Atom_ref.x (the reference program, unmodified sources, LAPACK 3.12): Enl.dat"))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("%-14s nfun=%5d channels=%3d  E1=%.15g  wall %.0f s" % (name, nfun, E.shape[0], E[0, 0], wall), flush=True)

def run_handoff():
    """the hand-off files.  (1) Enl.dat and CSs/MatElem_All.dat written by the PRODUCT's writers
    (bspatom_amd/host.py) are read by the REFERENCE'S OWN reader READ_COUP (ReadInputs.f90:277-369, unmodified, through
    oracle/ref/ref_handoff_driver.f90); what it read goes into the fixture next to the file texts.  (2) The same couplings
    written with the reference's two WRITE forms for MatElem_All.dat (PhotoIon.f90:255-266) by this Fortran runtime.  (3) The
    MatElem_All.dat the reference's own TRANS_AMP writes for KIND_PI = 3 in this container (header only: see the driver)."""
    sys.path.insert(0, ROOT)
    from bspatom_amd import host
    hx = os.path.join(ROOT, "oracle", "_ref", "ref_handoff.x")
    if not os.path.exists(hx):
        sys.exit("oracle/_ref/ref_handoff.x missing: run oracle/ref/build_ref.sh first")
    rng = np.random.default_rng(20261004)
    out = {}
    for kind_pi, nfields in ((3, 1), (5, 2), (8, 5)):
        lmax, nfun, n1_max = 1, 9, 3
        E = np.sort(rng.standard_normal((lmax + 1, nfun)) * 10.0 ** rng.integers(-9, 4, size=(lmax + 1, nfun)), axis=1)
        E[:, 0] = -0.5 / np.arange(1, lmax + 2) ** 2
        nbra = nket = n1_max * 2
        zT = (rng.standard_normal((nbra, nket, nfields)) + 1j * rng.standard_normal((nbra, nket, nfields))) * \
            10.0 ** rng.integers(-12, 6, size=(nbra, nket, nfields))
        zT[0, 1, 0] = 0.0
        zT[1, 2, 0] = 1.0 - 0.1j
        emax_fin = 0.25
        with tempfile.TemporaryDirectory(prefix="bspgold.") as tmp:
            rr = np.linspace(0.0, 1.0, 3)
            host.write_structure_outputs(nfun, lmax, E, 0, (rr, rr), tmp)
            host.write_matelem_all(os.path.join(tmp, "CSs", "MatElem_All.dat"), n1_max, zT)
            enl_text = open(os.path.join(tmp, "Enl.dat")).read()
            me_text = open(os.path.join(tmp, "CSs", "MatElem_All.dat")).read()
            ctl = "'R'\n%d %d %.17g %d %d %d\n" % (lmax, kind_pi, emax_fin, 1, 0, 0)
            p = subprocess.run([hx], input=ctl, cwd=tmp, capture_output=True, text=True)
            if p.returncode != 0 or "Error" in p.stdout:
                sys.exit("READ_COUP failed on the product's files: %s %s" % (p.stdout[-500:], p.stderr[-500:]))
            raw = open(os.path.join(tmp, "ref_handoff.bin"), "rb").read()
            hdr = np.frombuffer(raw, dtype=np.int32, count=6); off = 24
            rnfun, rn1, rnbra, rnket, rnf, rlmax = (int(x) for x in hdr)
            rE = np.frombuffer(raw, dtype=np.float64, count=rnfun * (rlmax + 1), offset=off).reshape(rlmax + 1, rnfun); off += 8 * rE.size
            rn01 = np.frombuffer(raw, dtype=np.int32, count=3 * (rlmax + 1), offset=off).reshape(3, rlmax + 1).T; off += 4 * rn01.size
            rz = np.frombuffer(raw, dtype=np.complex128, count=rnbra * rnket * rnf, offset=off).reshape(rnf, rnket, rnbra).transpose(2, 1, 0)
            # (2) the reference's WRITE forms through this Fortran runtime
            vals = "".join("%.17g %.17g\n" % (zT[ib, jk, i].real, zT[ib, jk, i].imag) for ib in range(nbra) for jk in range(ib, nket)
                           for i in range(nfields))
            p2 = subprocess.run([hx], input="'W'\n%d %d %d %d\n%s" % (n1_max, nbra, nket, nfields, vals), cwd=tmp, capture_output=True, text=True)
            if p2.returncode != 0:
                sys.exit("format driver failed: " + p2.stderr[-500:])
            me_fortran = open(os.path.join(tmp, "MatElem_All.dat")).read()
        tag = "pi%d_" % kind_pi
        out.update({tag + "E": E, tag + "zT": zT, tag + "emax_fin": emax_fin, tag + "n1_max": n1_max, tag + "nfields": nfields,
                    tag + "enl_text": np.array(enl_text), tag + "matelem_text": np.array(me_text),
                    tag + "matelem_text_fortran": np.array(me_fortran),
                    tag + "read_header": np.array([rnfun, rn1, rnbra, rnket, rnf, rlmax]), tag + "read_E": rE, tag + "read_n01": rn01,
                    tag + "read_z": rz})
        print("handoff KIND_PI=%d: READ_COUP read nfun=%d n1_max=%d nbra=%d nket=%d nfields=%d; max|dE| %.1e max|dz| rel %.1e; "
              "Fortran-written MatElem_All.dat %s the product's" % (kind_pi, rnfun, rn1, rnbra, rnket, rnf, np.max(np.abs(rE - E)),
              np.max(np.abs(np.triu(rz[:, :, 0] - zT[:, :, 0])) / np.max(np.abs(zT))), "==" if me_fortran == me_text else "!="))
    # (3) the reference's TRANS_AMP for KIND_PI = 3 as far as it runs here: header of its MatElem_All.dat
    with tempfile.TemporaryDirectory(prefix="bspgold.") as tmp:
        text = PI3["pi3_emax1"]
        os.makedirs(os.path.join(tmp, "CSs"))
        p = subprocess.run([REFX], input=text + "\n", cwd=tmp, capture_output=True, text=True, env=dict(os.environ, REF_DUMP_TRANS_AMP_PI3="1"))
        f = os.path.join(tmp, "CSs", "MatElem_All.dat")
        out["ref_pi3_matelem_text"] = np.array(open(f).read() if os.path.exists(f) else "")
        print("reference TRANS_AMP (KIND_PI = 3, MAKE_F_ANG not callable here) wrote MatElem_All.dat: %r" % str(out["ref_pi3_matelem_text"]))
    out["source"] = np.array("tests/golden/make_golden.py --handoff: oracle/_ref/ref_handoff.x (the reference's READ_COUP on the product's files; "
                             "the reference's WRITE forms through flang's runtime); oracle/_ref/ref_dump.x for the KIND_PI = 3 header")
    np.savez_compressed(os.path.join(HERE, "handoff.npz"), **out)


def run_vectors(name="c2_2048"):
    """Eigenvectors of the reference's DSYGV(1,'V','U') call (matrices.f90:248) at n = 2048, channel l = 0, as data: the matrices are
    the oracle's bands (bit-identical to the reference's Sij and Tij + Uij + Vij, tests/test_oracle_golden.py), the routine is
    LAPACK 3.12 DSYGV of the library the compiled reference is linked to (scipy's OpenBLAS, oracle/ref/lapack_forward.c) -- the
    call the reference makes, whose result SOLVE_SYSTEM frees before it returns (matrices.f90:386).  Stored: all eigenvalues; the
    96 eigenvector columns around the eigenvalue nearest zero (the block's edges moved to the widest gaps nearby), where
    neighbouring eigenvalues are closest and single vectors are determined worst; 64 columns spread over the spectrum."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle as orc
    from bspatom_amd.namelist import read_namelists
    nl = read_namelists(open(os.path.join(HERE, "inputs", name + ".inp")).read())
    c = orc.make_cfg(**{**nl["vars_bsp"], **nl["vars_tise"]})
    rt, aind, xg, wg = orc.grid(c)
    SB, HB = orc.assemble_bands(c, rt, aind, xg, wg, 0, 1)
    t0 = time.time()
    w, Z, info = orc.dsygv(orc.band_to_dense_upper(HB[0]), orc.band_to_dense_upper(SB), vectors=True)
    assert info == 0
    g = np.load(os.path.join(HERE, name + ".npz"))
    assert np.max(np.abs(w - g["E"][0])) <= 1e-13 * np.max(np.abs(w)), "not the spectrum the reference program wrote"
    n = len(w)
    z0 = int(np.argmin(np.abs(w)))
    gaps = np.diff(w)
    lo = max(z0 - 48, 8); hi = min(z0 + 48, n - 8)
    lo = lo - 8 + int(np.argmax(gaps[lo - 8:lo]))  + 1          # block = columns lo .. hi-1, edges at the widest gaps within 8
    hi = hi + int(np.argmax(gaps[hi - 1:hi + 7])) + 0
    block = np.arange(lo, hi + 1)
    sample = np.setdiff1d(np.unique(np.linspace(0, n - 1, 64).astype(int)), block)
    np.savez_compressed(os.path.join(HERE, "vec_" + name + ".npz"), w=w, block_idx=block, Zblock=Z[:, block], sample_idx=sample,
                        Zsample=Z[:, sample], gap_below=np.array(gaps[lo - 1]), gap_above=np.array(gaps[hi]),
                        source=np.array("scipy.linalg.lapack.dsygv (OpenBLAS 0.3.28 / LAPACK 3.12.0) on the oracle's bit-exact bands; "
                                        "%.0f s" % (time.time() - t0)))
    print("vec_%s: n = %d, block %d..%d (gaps at its edges %.2e / %.2e, smallest inside %.2e), %d sample columns"
          % (name, n, lo, hi, gaps[lo - 1], gaps[hi], gaps[lo:hi].min(), len(sample)))


def main():
    if "--handoff" in sys.argv[1:]:
        run_handoff()
        return
    if "--vectors" in sys.argv[1:]:
        run_vectors()
        return
    if not os.path.exists(REFX):
        sys.exit("oracle/_ref/ref_dump.x missing: run oracle/ref/build_ref.sh first")
    args = sys.argv[1:]
    cases = dict(CASES)
    if "--big" in args:
        cases.update(BIG); args.remove("--big")
    if "--dipole" in args:
        args.remove("--dipole")
        for name in (args or list(DIPOLE)):
            run_dipole(name, *DIPOLE[name])
        return
    if "--amp" in args:
        args.remove("--amp")
        for name in (args or list(AMP)):
            run_amp(name, AMP[name])
        return
    if "--spectra" in args:
        args.remove("--spectra")
        for name in (args or list(SPECTRA)):
            run_spectra(name, SPECTRA[name])
        return
    if "--pi3" in args:
        args.remove("--pi3")
        for name in (args or list(PI3)):
            run_pi3(name, PI3[name])
        return
    sel = args or list(cases)
    for name in sel:
        if name in PI3:
            run_pi3(name, PI3[name])
            continue
        if name in DIPOLE:
            run_dipole(name, *DIPOLE[name])
            continue
        text, mats = {**CASES, **BIG}[name]
        run_case(name, text, mats)
    man = {
        "generator": "tests/golden/make_golden.py",
        "reference": "carlosmwh1985/BspAtom @ /root/reference (unmodified sources; INQUIRE(DIRECTORY=) spelled INQUIRE(FILE=) for flang)",
        "compiler": subprocess.run(["/opt/rocm/lib/llvm/bin/flang", "--version"], capture_output=True, text=True).stdout.split("\n")[0],
        "flags": "-O2 (x86-64 baseline, no FMA)",
        "lapack": "scipy-bundled OpenBLAS 0.3.28 / LAPACK 3.12.0 (scipy_dsygv_), 8 threads",
    }
    mpath = os.path.join(HERE, "MANIFEST.json")
    if os.path.exists(mpath):                      # keep the notes other modes left there
        old = json.load(open(mpath)); old.update(man); man = old
    json.dump(man, open(mpath, "w"), indent=1)

if __name__ == "__main__":
    main()
