"""Pins the CPU oracle (oracle/bsp_oracle.c) to fixtures produced by the compiled reference.

Bars: integers, knots, GL rule, Aind, S, T, V, U_l -- bit-for-bit; spectra -- LAPACK-vs-LAPACK
noise: normwise |dE|/lambda_max <= 1e-13 everywhere, relative <= 1e-10 on linear grids.
"""
import numpy as np
import pytest
from conftest import load_golden, golden_input, SMALL_CASES, ulp_diff
import oracle as orc
from bspatom_amd.namelist import read_namelists


def cfg_from_input(name):
    nl = read_namelists(open(golden_input(name)).read())
    kw = {}
    kw.update(nl.get("vars_bsp", {}))
    kw.update(nl.get("vars_tise", {}))
    return orc.make_cfg(**kw)


@pytest.mark.parametrize("name", SMALL_CASES + ["lin1024", "wf_fatal"])
def test_derived_sizes_knots_rule(name):
    g = load_golden(name)
    c = cfg_from_input(name)
    nfun, k, ka, nkp, nointv, nbc1, nbc2, lmax = [int(v) for v in g["sizes"]]
    assert (c.nfun, c.k, c.ka, c.nkp, c.nointv, c.nbc1, c.nbc2, c.lmax) == (nfun, k, ka, nkp, nointv, nbc1, nbc2, lmax)
    rt, aind, xg, wg = orc.grid(c)
    assert np.array_equal(rt, g["rt"])
    assert np.array_equal(xg, g["xg"]) and np.array_equal(wg, g["wg"])
    assert np.array_equal(aind, g["aind"])


@pytest.mark.parametrize("name", [n for n in SMALL_CASES if n not in ("lin256", "yuk256")])
def test_matrix_svt_bit_exact(name):
    """Reference-faithful per-pair loop (orc_matrix_svt) against the reference's dense matrices."""
    g = load_golden(name)
    c = cfg_from_input(name)
    rt, aind, xg, wg = orc.grid(c)
    S, V, T, U = orc.matrix_svt(c, rt, aind, xg, wg)
    k = c.k
    for d in range(k):
        n = c.nfun - d
        assert np.array_equal(np.diagonal(S, d), g["Sb"][d, :n]), "S diag %d" % d
        assert np.array_equal(np.diagonal(T, d), g["Tb"][d, :n]), "T diag %d" % d
        assert np.array_equal(np.diagonal(V, d), g["Vb"][d, :n]), "V diag %d" % d
        for l in range(c.lmax + 1):
            assert np.array_equal(np.diagonal(U[l], d), g["Ub"][l, d, :n]), "U l=%d diag %d" % (l, d)
    # exact zeros outside the band, like the reference
    mask = np.abs(np.subtract.outer(np.arange(c.nfun), np.arange(c.nfun))) >= k
    assert not np.any(S[mask]) and not np.any(T[mask]) and not np.any(V[mask])


@pytest.mark.parametrize("name", SMALL_CASES)
def test_band_assembly_bit_exact(name):
    """Interval-table band form (orc_assemble_bands): H_l = (T+U_l)+V and S, bit-for-bit."""
    g = load_golden(name)
    c = cfg_from_input(name)
    rt, aind, xg, wg = orc.grid(c)
    SB, HB = orc.assemble_bands(c, rt, aind, xg, wg)
    assert np.array_equal(SB, g["Sb"])
    for l in range(c.lmax + 1):
        Href = (g["Tb"] + g["Ub"][l]) + g["Vb"]
        assert np.array_equal(HB[l], Href), "H band l=%d" % l


def spectrum_errors(E, Eref):
    lam = np.max(np.abs(Eref))
    rel = np.max(np.abs(E - Eref) / np.abs(Eref))
    nrm = np.max(np.abs(E - Eref)) / lam
    return rel, nrm


@pytest.mark.parametrize("name", SMALL_CASES)
def test_spectra_lapack(name):
    g = load_golden(name)
    c = cfg_from_input(name)
    E, vec, _ = orc.solve_all(c)
    lin = (c.kind_grid == 0)
    for l in range(c.lmax + 1):
        rel, nrm = spectrum_errors(E[l], g["E"][l])
        assert nrm <= 1e-13, (name, l, nrm)
        if lin:
            assert rel <= 1e-10, (name, l, rel)


@pytest.mark.parametrize("name", ["c1_exp", "c1_lin", "bc1", "ka_ra"])
def test_spectra_plain_c_chain(name):
    """The textbook C chain (orc_dsygv) agrees with LAPACK normwise."""
    g = load_golden(name)
    c = cfg_from_input(name)
    rt, aind, xg, wg = orc.grid(c)
    SB, HB = orc.assemble_bands(c, rt, aind, xg, wg)
    w, v, info = orc.dsygv(orc.band_to_dense_upper(HB[0]), orc.band_to_dense_upper(SB), impl="c")
    assert info == 0
    rel, nrm = spectrum_errors(w, g["E"][0])
    assert nrm <= 5e-13, (name, nrm)
    # S-orthonormal eigenvectors
    S = orc.band_to_dense_upper(SB); S = S + np.triu(S, 1).T
    assert np.max(np.abs(v.T @ S @ v - np.eye(c.nfun))) < 1e-9


@pytest.mark.parametrize("name", SMALL_CASES)
def test_write_wf(name):
    g = load_golden(name)
    c = cfg_from_input(name)
    E, vec, (rt, aind, xg, wg) = orc.solve_all(c)
    r, u = orc.write_wf(c, rt, vec)
    rows = g["wf_rows"]; idx = g["wf_idx"]
    # '(2G20.10)' keeps 10 significant digits
    assert np.allclose(r[idx], rows[:, 0], rtol=2e-10, atol=1e-300)
    sgn = np.sign(np.dot(u[idx], rows[:, 1]))
    scale = np.max(np.abs(rows[:, 1]))
    assert np.max(np.abs(sgn * u[idx] - rows[:, 1])) <= 2e-8 * scale


def test_wf_fatal_edge_case():
    """ra=0.5, rb=40: the last WRITE_WF point rounds above rb -> interv left=1 -> BSPLVB STOP."""
    g = load_golden("wf_fatal")
    assert int(g["fatal"][0]) == 1 and g["E"].shape[0] == 1     # reference stopped after l = l_ini
    c = cfg_from_input("wf_fatal")
    rt, aind, xg, wg = orc.grid(c)
    with pytest.raises(RuntimeError, match="BSPLVB"):
        orc.write_wf(c, rt, np.ones(c.nfun))


def test_rydberg_known_answers():
    """oracle reproduces the survey's measured reference values."""
    g = load_golden("bsp0")
    assert abs(g["E"][0, 0] - (-0.499999999965076)) < 1e-14
    assert abs(g["E"][1, 0] - (-0.124999999944288)) < 1e-14
    assert abs(g["E"][2, 0] - (-0.0555555554481607)) < 1e-14
    g = load_golden("c1_exp")
    assert abs(g["E"][0, 0] - (-0.499999999999882)) < 1e-14


@pytest.mark.parametrize("name", ["dip_len_lin", "dip_vel_lin", "dip_len_exp", "dip_vel_exp"])
def test_oracle_dipole_bands_bit_exact(name):
    """the oracle's restatement of the dipole sums against the compiled reference's rij."""
    import oracle as orc
    from bspatom_amd.namelist import read_namelists
    g = load_golden(name)
    nl = read_namelists(str(g["namelist"]))
    kw = {}
    kw.update(nl["vars_bsp"]); kw.update(nl["vars_tise"])
    c = orc.make_cfg(**kw)
    rt, aind, xg, wg = orc.grid(c)
    RB = orc.dipole_bands(c, rt, aind, xg, wg)
    if int(g["kind_pi"][0]) == 1:
        assert np.array_equal(RB[0], g["r1f"])
    else:
        assert np.array_equal(RB[1], g["r1f"]) and np.array_equal(RB[2], g["r2f"])


AMP_CASES = ["ta_len_s", "ta_vel_s", "ta_len_p", "ta_vel_p"]


@pytest.mark.parametrize("name", AMP_CASES)
def test_oracle_trans_amp_vs_reference(name):
    """TRANS_AMP for KIND_PI = 1, 2 (PhotoIon.f90:50-107) restated: on the reference's own rij, eigenvectors and
    spectrum the amplitudes T_fi(n0_fin:n1_fin) must come out to rounding (DGEMV / DDOT summation order is the BLAS
    library's: 1e-13 of max|T|), the final-state window (matrices.f90:272-283) exactly."""
    import oracle as orc
    g = load_golden(name)
    nfun, kp, n0i, l0, m0, lf, mf, mph, n0f, n1f = (int(v) for v in g["head"])
    a, b, _ = orc.final_state_limits(g["E_fin"], float(g["emax_fin"][0]))
    assert (a, b) == (n0f, n1f)
    T = orc.trans_amp(kp, l0, m0, lf, mf, mph, g["r1"], g["r2"], g["ci_ini"], g["ci_fin"], g["E_fin"], n0f, n1f)
    assert np.max(np.abs(T - g["T_fi"])) <= 1e-13 * np.max(np.abs(g["T_fi"]))


def test_three_j_known_values():
    """THREE_J restatement against closed forms: (l 1 l+1; 0 0 0)^2 = (l+1)/((2l+1)(2l+3)), (1 1 0; 0 0 0) = -1/sqrt 3,
    selection rules, and the orthogonality sum over m."""
    import oracle as orc
    assert abs(orc.three_j(1, 1, 0, 0, 0, 0) + 1.0 / np.sqrt(3.0)) < 1e-15
    for l in range(0, 8):
        v = orc.three_j(l + 1, 1, l, 0, 0, 0)
        assert abs(v * v - (l + 1.0) / ((2 * l + 1.0) * (2 * l + 3.0))) < 1e-14
    assert orc.three_j(2, 1, 0, 0, 0, 0) == 0.0 and orc.three_j(1, 1, 1, 0, 0, 0) == 0.0 and orc.three_j(1, 1, 1, 1, 0, 0) == 0.0
    for j3 in (1, 2, 3):
        s = sum(orc.three_j(2, 1, j3, m1, m2, -m1 - m2) ** 2 for m1 in range(-2, 3) for m2 in (-1, 0, 1) if abs(m1 + m2) <= j3)
        assert abs(s - 1.0) < 1e-13


# ---- 113-bit truth fixtures (tests/golden/make_truth.py) ---------------------------------------------------------
@pytest.mark.parametrize("name", ["c1_lin", "lin256"])
def test_truth_fixture_reproduces_and_brackets_the_reference(name):
    """The committed truth values are what oracle/truth_quad.c computes from the oracle's bands (bit for bit: the
    computation is deterministic), every one lies within LAPACK's normwise error of the reference's value, and a
    dense double-precision solve of the same pencil (scipy eigh, another LAPACK driver) agrees to 1e-13 normwise."""
    import scipy.linalg as sla
    from oracle import truth as qt
    from tests_truth import case_cfg
    t = load_golden("truth_" + name); g = load_golden(name)
    c = case_cfg(name)
    rt, aind, xg, wg = orc.grid(c)
    nch = g["E"].shape[0]
    SB, HB = orc.assemble_bands(c, rt, aind, xg, wg, 0, nch)
    for l in (0, nch - 1):
        sel = t["chan"] == l
        idx = t["idx"][sel]
        lam = float(np.max(np.abs(g["E"][l])))
        hi, lo = qt.band_eigs(SB, HB[l], idx, g["E"][l][idx], lam)
        assert np.array_equal(hi, t["hi"][sel]) and np.array_equal(lo, t["lo"][sel])
        assert np.max(np.abs(g["E"][l][idx] - hi)) <= 1e-13 * lam
        S = orc.band_to_dense_upper(SB); S = S + np.triu(S, 1).T
        H = orc.band_to_dense_upper(HB[l]); H = H + np.triu(H, 1).T
        w = sla.eigh(H, S, eigvals_only=True)
        assert np.max(np.abs(w[idx] - hi)) <= 1e-13 * lam


def test_truth_count_is_an_inertia():
    """orc_truth_count at a point between two well-separated eigenvalues returns the index of the upper one."""
    from oracle import truth as qt
    from tests_truth import case_cfg
    g = load_golden("c1_lin")
    c = case_cfg("c1_lin")
    rt, aind, xg, wg = orc.grid(c)
    SB, HB = orc.assemble_bands(c, rt, aind, xg, wg, 0, 1)
    E = g["E"][0]
    for m in (1, 5, 30, 63):
        x = 0.5 * (E[m - 1] + E[m])
        cnt = qt.lib().orc_truth_count(SB.shape[1], SB.shape[0], qt._p(np.ascontiguousarray(SB)),
                                       qt._p(np.ascontiguousarray(HB[0])), float(x), 0.0)
        assert cnt == m
