"""GPU parity tests of the whole hot path through the C ABI against the golden fixtures generated
from the compiled reference (LAPACK 3.12 DSYGV) -- SURVEY 8(d) figures:
  (i)  per-eigenvalue relative error  max_i |dE_i|/|E_ref,i|      <= 1e-10 on linear grids
  (ii) normwise                        max_i |dE_i|/lambda_max      <= 1e-13 on every grid
  (iii) Rydberg check for well-contained Coulomb states."""
import os
import numpy as np
import pytest
from conftest import load_golden, golden_input, SMALL_CASES, ROOT
from test_gpu_stages import input_from_case, note

pytestmark = pytest.mark.gpu
from bspatom_amd import capi


def figures(E, Eref):
    lam = np.max(np.abs(Eref))
    return np.max(np.abs(E - Eref) / np.abs(Eref)), np.max(np.abs(E - Eref)) / lam


def full_size_bar(E, Eref, tag):
    """Linear-grid parity bar:  |dE| <= 1e-10 |E| + 1/2 eps lambda_max  for EVERY eigenvalue, i.e. 1e-10 relative
    (north_star) except where that would be finer than half an ulp of the matrix norm (eigenvalues within
    ~2e-3 of zero at lambda_max ~ 1.5e3).  There the reference's own LAPACK rounding is 1e-13..3e-13 absolute
    (measured against the exact Rydberg values), so no solver can reproduce it to 1e-10 relative.
    Also: normwise 1e-13, and the exceptions to the pure relative bar are counted and must all be such
    near-zero eigenvalues."""
    eps = np.finfo(float).eps
    lam = np.max(np.abs(Eref))
    d = np.abs(E - Eref)
    rel = d / np.abs(Eref)
    exc = rel > 1e-10
    note("%s: worst rel %.2e at E=%.2e (|dE| %.1e = %.2f eps*lam_max)  normwise %.2e  exceptions to pure 1e-10 relative: %d"
         % (tag, np.max(rel), Eref[np.argmax(rel)], d[np.argmax(rel)], d[np.argmax(rel)] / (eps * lam), np.max(d) / lam,
            int(np.sum(exc))))
    assert np.all(d <= 1e-10 * np.abs(Eref) + 0.5 * eps * lam)
    assert np.max(d) / lam <= 1e-13
    assert np.all(np.abs(Eref[exc]) < 0.5 * eps * lam / 1e-10)


@pytest.mark.parametrize("name", SMALL_CASES + ["lin1024", "c2_2048", "c3_1024_l31", "c5_1024_k11"])
def test_spectra_vs_reference(name):
    g = load_golden(name)
    inp = input_from_case(name)
    prob = capi.Problem(inp)
    lmax = prob.lmax
    E, info = prob.solve(0, lmax + 1)
    assert np.all(info == 0)
    lin = inp.kind_grid == 0
    for l in range(lmax + 1):
        rel, nrm = figures(E[l], g["E"][l])
        note("solve %s l=%d n=%d: rel %.2e normwise %.2e  timing %s" % (name, l, prob.nfun, rel, nrm, prob.last_timing()))
        assert nrm <= 1e-13
        if lin:
            # relative 1e-10 wherever |E| >= 1e-3; eigenvalues nearer to zero are compared with the mixed bound
            # (the reference's own rounding is ~1e-13 absolute there, see full_size_bar)
            full_size_bar(E[l], g["E"][l], "  bar %s l=%d" % (name, l))
    prob.close()


def test_rydberg_series():
    """Hydrogen, linear grid, rb=400, n=2048 (BASELINE config C2): E_n = -1/(2 n^2), n <= 8."""
    prob = capi.Problem(input_from_case("c2_2048"))
    E, info = prob.solve(0, 1)
    for nq in range(1, 9):
        exact = -0.5 / nq ** 2
        assert abs(E[0, nq - 1] - exact) / abs(exact) < 1e-10
    prob.close()


def test_channel_offset_and_batch():
    """Channels l0..l0+nl-1 solved as one batch equal the same channels solved one by one."""
    inp = input_from_case("lin256", l_fin=0)
    prob = capi.Problem(inp)
    Eb, info = prob.solve(2, 5)
    for i, l in enumerate(range(2, 7)):
        E1, _ = prob.solve(l, 1)
        assert np.array_equal(E1[0], Eb[i])
    prob.close()


@pytest.mark.parametrize("name", SMALL_CASES)
def test_eigvec_and_wf(name):
    g = load_golden(name)
    inp = input_from_case(name)
    prob = capi.Problem(inp)
    E, info = prob.solve(0, prob.lmax + 1)
    c = prob.eigvec(inp.l_ini, inp.n0_ini)
    SB, HB = prob.assemble(inp.l_ini, 1)
    # residual of the banded pencil and S-normalisation
    n, k = prob.nfun, prob.k
    def bmv(Bd, x):
        y = Bd[0] * x
        for d in range(1, k):
            y[:n - d] += Bd[d, :n - d] * x[d:]
            y[d:] += Bd[d, :n - d] * x[:n - d]
        return y
    Sx = bmv(SB, c); Hx = bmv(HB[0], c)
    lam = E[inp.l_ini, inp.n0_ini - 1]
    assert abs(c @ Sx - 1.0) < 1e-12
    res = np.max(np.abs(Hx - lam * Sx)) / np.max(np.abs(g["E"]))
    note("eigvec %s residual/lambda_max %.2e" % (name, res))
    assert res < 1e-12
    r, u = prob.write_wf(c)
    rows = g["wf_rows"]; idx = g["wf_idx"]
    assert np.allclose(r[idx], rows[:, 0], rtol=2e-10, atol=1e-300)
    sgn = np.sign(np.dot(u[idx], rows[:, 1]))
    scale = np.max(np.abs(rows[:, 1]))
    err = np.max(np.abs(sgn * u[idx] - rows[:, 1])) / scale
    note("wf %s max err / max|u| %.2e" % (name, err))
    assert err <= 2e-8          # '(2G20.10)' keeps 10 digits; eigenvector itself agrees to ~1e-10
    prob.close()


def test_wf_fatal_edge_case():
    """Reference STOPs in BSPLVB when the last tabulation point rounds above rb: same status here."""
    prob = capi.Problem(input_from_case("wf_fatal"))
    with pytest.raises(capi.BspAtomError) as ei:
        prob.write_wf(np.ones(prob.nfun))
    assert ei.value.code == -3
    prob.close()


def test_overlap_not_positive_definite():
    """DSYGV info = n + i when S is not PD (matrices.f90:250-254 prints and STOPs)."""
    g = load_golden("c1_lin")
    k = int(g["sizes"][1]); n = int(g["sizes"][0])
    S = np.zeros((n, n)); H = np.zeros((n, n))
    for d in range(k):
        idx = np.arange(n - d)
        S[idx, idx + d] = g["Sb"][d, :n - d]; H[idx, idx + d] = ((g["Tb"] + g["Ub"][0]) + g["Vb"])[d, :n - d]
    S[9, 9] = -1.0
    w, Z, U, info = capi.dsygv(H, S, jobz="N", uplo="U")
    assert info == n + 10


@pytest.mark.parametrize("name", ["c1_lin", "lin256"])
def test_bsp_dsygv_symbol(name):
    """bsp_dsygv_ with the reference's call shape DSYGV(1,'V','U',...)."""
    g = load_golden(name)
    k = int(g["sizes"][1]); n = int(g["sizes"][0])
    S = np.zeros((n, n)); H = np.zeros((n, n))
    for d in range(k):
        idx = np.arange(n - d)
        S[idx, idx + d] = g["Sb"][d, :n - d]; H[idx, idx + d] = ((g["Tb"] + g["Ub"][0]) + g["Vb"])[d, :n - d]
    w, Z, U, info = capi.dsygv(H, S, jobz="V", uplo="U")
    assert info == 0
    rel, nrm = figures(w, g["E"][0])
    note("bsp_dsygv_ %s rel %.2e normwise %.2e" % (name, rel, nrm))
    assert nrm <= 1e-13 and rel <= 1e-10
    Sf = S + np.triu(S, 1).T; Hf = H + np.triu(H, 1).T
    orth = np.max(np.abs(Z.T @ Sf @ Z - np.eye(n)))
    resid = np.max(np.abs(Hf @ Z - Sf @ Z * w)) / np.max(np.abs(w))
    note("bsp_dsygv_ %s Z^T S Z - I %.2e  residual %.2e" % (name, orth, resid))
    assert orth < 1e-8 and resid < 1e-11
    assert np.max(np.abs(np.triu(U).T @ np.triu(U) - Sf)) < 1e-13


def _read_enl(path, nfun, nch):
    lines = open(path).read().split("\n")
    assert int(lines[0]) == nfun
    vals = [float(l.split()[1]) for l in lines[1:] if l.strip()]
    return np.array(vals).reshape(nch, nfun)


@pytest.mark.parametrize("name", ["bsp0", "c1_lin", "rogers"])
def test_fortran_host_drop_in(tmp_path, name):
    """bsp_atom_host.x < input: stdout eigenvalue lines, Enl.dat and wf_n0.dat against the reference's."""
    import subprocess, re
    exe = os.path.join(ROOT, "bspatom_amd", "bsp_atom_host.x")
    if not os.path.exists(exe):
        pytest.skip("Fortran host not built (no flang)")
    g = load_golden(name)
    with open(golden_input(name)) as fin:
        p = subprocess.run([exe], stdin=fin, cwd=tmp_path, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "Program Finished!" in p.stdout
    nch, nfun = g["E"].shape
    E = _read_enl(tmp_path / "Enl.dat", nfun, nch)
    lam = np.max(np.abs(g["E"]))
    assert np.max(np.abs(E - g["E"])) / lam < 1e-13
    wf = np.loadtxt(tmp_path / "wf_n0.dat")
    assert wf.shape == (10001, 2)
    rows = g["wf_rows"]; idx = g["wf_idx"]
    sgn = np.sign(np.dot(wf[idx, 1], rows[:, 1]))
    assert np.max(np.abs(sgn * wf[idx, 1] - rows[:, 1])) <= 2e-8 * np.max(np.abs(rows[:, 1]))
    # the first-20 table of every channel carries the reference's labels i+l
    mine = [l for l in p.stdout.split("\n") if re.match(r"^\s+\d+\s+-?\d?\.\d", l)]
    ref = [l for l in str(g["stdout"]).split("\n") if re.match(r"^\s+\d+\s+-?\d?\.\d", l)]
    assert len(mine) == len(ref)
    assert [int(a.split()[0]) for a in mine] == [int(b.split()[0]) for b in ref]


def test_python_host_outputs(tmp_path):
    from bspatom_amd import host
    g = load_golden("c1_lin")
    E, c, text = host.run(open(golden_input("c1_lin")).read(), outdir=str(tmp_path))
    nch, nfun = g["E"].shape
    Ef = _read_enl(tmp_path / "Enl.dat", nfun, nch)
    assert np.max(np.abs(Ef - g["E"])) / np.max(np.abs(g["E"])) < 1e-13
    wf = np.loadtxt(tmp_path / "wf_n0.dat")
    assert wf.shape == (10001, 2)
    assert "Program Finished!" in text


def test_c4_channels_at_full_size():
    """BASELINE configs[3] size (n=4096, k=9, rb=800): channels l=0,1 against the reference's spectra."""
    g = load_golden("c4_4096")
    prob = capi.Problem(input_from_case("c4_4096"))
    E, info = prob.solve(0, 2)
    assert np.all(info == 0)
    for l in range(2):
        full_size_bar(E[l], g["E"][l], "solve c4_4096 l=%d" % l)
        # truth check: the GPU spectrum is at least as close to the exact Rydberg values as the reference's
        nq = np.arange(1, 11) + l
        exact = -0.5 / nq ** 2
        assert np.max(np.abs(E[l, :10] - exact)) <= np.max(np.abs(g["E"][l, :10] - exact)) + 1e-13
    prob.close()


def test_full_size_properties_128_channels():
    """At BASELINE's full batch (128 channels, n=4096) only size-independent properties are affordable:
    every spectrum sorted, info 0, Rydberg series of l=0..3, and interlacing-like monotonicity in l of the
    lowest eigenvalue (the centrifugal term is positive), plus equality with the 2-channel reference run."""
    g = load_golden("c4_4096")
    prob = capi.Problem(input_from_case("c4_4096", l_fin=127))
    E, info = prob.solve(0, 128)
    assert np.all(info == 0)
    assert np.all(np.diff(E, axis=1) >= 0)
    for l in range(4):
        nq = np.arange(l + 1, l + 5)
        assert np.max(np.abs(E[l, :4] + 0.5 / nq ** 2) * 2 * nq ** 2) < 1e-9
    assert np.all(np.diff(E[:, 0]) > 0)
    for l in range(2):
        full_size_bar(E[l], g["E"][l], "batch128 c4_4096 l=%d" % l)
    prob.close()


def test_full_size_batch_is_deterministic():
    """The paired bulge-chasing workgroups synchronise through published progress words; the arithmetic and its
    order do not depend on their timing, so repeated solves must agree bit for bit (a race shows up as a difference)."""
    prob = capi.Problem(input_from_case("c4_4096", l_fin=127))
    E0, info = prob.solve(0, 128)
    assert np.all(info == 0)
    for _ in range(3):
        E, info = prob.solve(0, 128)
        assert np.array_equal(E, E0)
    prob.close()


def test_invalid_requests():
    prob = capi.Problem(input_from_case("c1_lin"))
    with pytest.raises(capi.BspAtomError):
        prob.solve(0, 0)                        # no channels
    with pytest.raises(capi.BspAtomError):
        prob.eigvec(0, 1)                       # no solve yet covering l=0
    prob.solve(0, 2)
    with pytest.raises(capi.BspAtomError):
        prob.eigvec(5, 1)                       # channel outside the last solve
    with pytest.raises(capi.BspAtomError):
        prob.eigvec(0, 0)                       # n0 is 1-based
    prob.close()


# ---- the eigenvector block the KIND_PI >= 3 branch keeps, and Eigenvec_All.dat -----------------
@pytest.mark.gpu
@pytest.mark.parametrize("name", ["c1_lin", "bsp0", "rogers", "lin256"])
def test_eigvecs_block_vs_lapack(name):
    """Columns 1..nvec of DSYGV's 'V' output per channel (matrices.f90:248, kept as ctemp(:,1:ntemp,l) at :331).
    Reference: LAPACK dsygv('V','U') -- the library the compiled reference links -- on the reference's own S, H_l
    (golden fixtures).  Eigenvectors are defined up to sign: |c_gpu^T S c_ref| = 1, S-orthonormality, residual."""
    import oracle as orc
    g = load_golden(name)
    inp = input_from_case(name)
    prob = capi.Problem(inp)
    nch, n = g["E"].shape
    E, info = prob.solve(0, nch)
    assert np.all(info == 0)
    k = g["Sb"].shape[0]
    nvec = min(n, 40)
    for l in range(nch):
        Su = orc.band_to_dense_upper(g["Sb"])
        Hu = orc.band_to_dense_upper((g["Tb"] + g["Ub"][l]) + g["Vb"])
        w, Zref, linfo = orc.dsygv(Hu, Su)
        assert linfo == 0
        S = Su + np.triu(Su, 1).T
        H = Hu + np.triu(Hu, 1).T
        Z = prob.eigvecs(l, 1, nvec)                                  # (nvec, n)
        G = Z @ S @ Z.T
        assert np.max(np.abs(G - np.eye(nvec))) < 1e-9
        lam = np.max(np.abs(w))
        for j in range(nvec):
            res = np.max(np.abs(H @ Z[j] - E[l, j] * (S @ Z[j]))) / lam
            assert res < 1e-12
            # overlap with the reference vector unless its eigenvalue has a neighbour closer than 1e-7 relative
            gap = min(abs(w[j] - w[j - 1]) if j > 0 else np.inf, abs(w[j + 1] - w[j]) if j + 1 < n else np.inf)
            if gap > 1e-7 * lam:
                ov = abs(Z[j] @ S @ Zref[:, j])
                assert abs(ov - 1.0) < 1e-7, (name, l, j, ov)
    note("eigvecs %s: %d channels x %d vectors, S-orthonormal to %.1e" % (name, nch, nvec, np.max(np.abs(G - np.eye(nvec)))))
    prob.close()


@pytest.mark.gpu
def test_eigenvec_all_file_round_trip(tmp_path):
    """Eigenvec_All.dat in the reference's layout (matrices.f90:366-378, FORMAT(I5,5000G20.10)) read back the way
    READ_EIGENVEC does (ReadInputs.f90:792-830): header, channel records, 10 significant digits."""
    from bspatom_amd import host
    prob = capi.Problem(input_from_case("c1_lin"))
    nch = load_golden("c1_lin")["E"].shape[0]
    prob.solve(0, nch)
    n1 = 12
    path = tmp_path / "Eigenvec_All.dat"
    host.write_eigenvec_all(str(path), prob, nch - 1, n1)
    nfun, n1r, lmax, c = host.read_eigenvec_all(str(path))
    assert (nfun, n1r, lmax) == (prob.nfun, n1, nch - 1)
    for l in range(nch):
        Z = prob.eigvecs(l, 1, n1)
        assert np.max(np.abs(c[l] - Z)) <= 1e-9 * np.max(np.abs(Z))
    first = open(path).readlines()[2]
    assert len(first.rstrip("\n")) == 5 + 20 * prob.nfun
    prob.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["pi3_emax1", "pi3_default", "pi5_emax05", "pi8_emax1", "pi3_nobound"])
def test_kind_pi3_limits_and_eigenvec_all_vs_reference(tmp_path, name):
    """`Bsp_Atom.x < input` with KIND_PI >= 3 up to the end of SOLVE_SYSTEM (matrices.f90:290-378) against the
    compiled reference: the 'NUMBER OF BOUND STATES' / 'LIMITS FOR l' / 'n1_max' lines character by character
    (they depend on the sign of eigenvalues and on comparisons with Emax_fin only), the Eigenvec_All.dat header,
    and every eigenvector record to the file's 10 digits up to the sign LAPACK happened to give it:
    |c_gpu -/+ c_ref| <= 2e-8 max|c| (the fixtures hold continuum states with gaps >= 1e-2; inverse iteration and
    DSTEQR agree to ~1e-11 there)."""
    from bspatom_amd import host
    g = load_golden(name)
    nfun, lmax, n1_max, kind_pi = (int(v) for v in g["sizes"])
    E, c, text = host.run(str(g["namelist"]), outdir=str(tmp_path))
    mine = [x for x in text.split("\n") if ("BOUND STATES" in x or "LIMITS FOR" in x or "n1_max" in x)]
    assert mine == [x.rstrip() for x in str(g["limits"]).split("\n")]
    lam = np.max(np.abs(g["E"]))
    assert np.max(np.abs(E - g["E"])) <= 1e-13 * lam
    lines = open(tmp_path / "Eigenvec_All.dat").read().split("\n")
    assert "\n".join(lines[:2]) == str(g["eva_head"])
    nf, n1, lm, C = host.read_eigenvec_all(str(tmp_path / "Eigenvec_All.dat"))
    assert (nf, n1, lm) == (nfun, n1_max, lmax)
    R = g["C"]
    worst = 0.0
    for l in range(lmax + 1):
        for ni in range(n1_max):
            sc = np.max(np.abs(R[l, ni]))
            d = min(np.max(np.abs(C[l, ni] - R[l, ni])), np.max(np.abs(C[l, ni] + R[l, ni]))) / sc
            worst = max(worst, d)
    assert worst <= 2e-8, worst
    assert "Program Finished!" not in text


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["pi3_emax1", "pi3_nobound", "pi5_emax05"])
def test_fortran_host_kind_pi3(tmp_path, name):
    """bsp_atom_host.x with KIND_PI >= 3: the same checks as the Python host above, through the Fortran binding of
    bspatom_eigvecs; here WRITE(80,*) and FORMAT(I5,5000G20.10) are flang's own, as in the reference build."""
    import subprocess
    from bspatom_amd import host
    exe = os.path.join(ROOT, "bspatom_amd", "bsp_atom_host.x")
    if not os.path.exists(exe):
        pytest.skip("Fortran host not built (no flang)")
    g = load_golden(name)
    nfun, lmax, n1_max, kind_pi = (int(v) for v in g["sizes"])
    with open(golden_input(name)) as fin:
        p = subprocess.run([exe], stdin=fin, cwd=tmp_path, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    mine = [x.rstrip() for x in p.stdout.split("\n") if ("BOUND STATES" in x or "LIMITS FOR" in x or "n1_max" in x)]
    assert mine == [x.rstrip() for x in str(g["limits"]).split("\n")]
    lines = open(tmp_path / "Eigenvec_All.dat").read().split("\n")
    assert "\n".join(lines[:2]) == str(g["eva_head"])
    assert len(lines[2]) == len(str(g["eva_row"]))
    nf, n1, lm, C = host.read_eigenvec_all(str(tmp_path / "Eigenvec_All.dat"))
    assert (nf, n1, lm) == (nfun, n1_max, lmax)
    R = g["C"]
    for l in range(lmax + 1):
        for ni in range(n1_max):
            sc = np.max(np.abs(R[l, ni]))
            d = min(np.max(np.abs(C[l, ni] - R[l, ni])), np.max(np.abs(C[l, ni] + R[l, ni]))) / sc
            assert d <= 2e-8, (l, ni, d)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["ta_len_s", "ta_vel_s", "ta_len_p", "ta_vel_p"])
def test_transition_amplitudes_vs_reference(name):
    """KIND_PI = 1 (length) and 2 (velocity) up to the end of TRANS_AMP (PhotoIon.f90:1-107) against the compiled
    reference (one diagnostic WRITE shortened, oracle/ref/build_ref.sh): the final-state window exactly; T_fi(n) for
    every final state INCLUDING its sign, after the reference's eigenvector signs (LAPACK's, arbitrary) are mapped to
    this library's convention (first coefficient above 1e-8 positive) -- T is bilinear in the two vectors.  Tolerance:
    1e-10 of max|T| on linear grids (measured 1e-13 .. 8e-13); 3e-7 on the grid with an exponential part, where two
    LAPACK runs already differ by 2e-8 .. 3e-7 in the eigenvalues next to zero and those enter through the
    density-of-states factor sqrt(2 / (E(n+1) - E(n-1))) (measured 1.7e-8)."""
    from bspatom_amd import host
    g = load_golden(name)
    nfun, kp, n0i, l0, m0, lf, mf, mph, n0f, n1f = (int(v) for v in g["head"])
    r = host.trans_amp(str(g["namelist"]))
    assert (r["l_fin"], r["m_fin"], r["n0_fin"], r["n1_fin"]) == (lf, mf, n0f, n1f)
    def conv_sign(c):                        # eigvec.hip: first coefficient above 1e-8 of the largest one is positive
        big = np.where(np.abs(c) > 1e-8 * np.max(np.abs(c)))[0]
        return 1.0 if (len(big) == 0 or c[big[0]] > 0) else -1.0
    si = conv_sign(g["ci_ini"])
    Tref = np.array([g["T_fi"][i] * si * conv_sign(g["ci_fin"][:, i]) for i in range(n1f - n0f + 1)])
    err = np.max(np.abs(r["T_fi"] - Tref)) / np.max(np.abs(Tref))
    note("trans_amp %s max|T| %.4g err %.2e" % (name, np.max(np.abs(Tref)), err))
    assert err <= (3e-7 if "KIND_GRID=1" in str(g["namelist"]) else 1e-10), err
    for line in r["stdout"].split("\n"):
        assert line in [x.strip() for x in str(g["lines"]).split("\n")], line
