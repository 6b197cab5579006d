"""Helpers shared by the truth-fixture tests: the oracle configuration of a golden input."""
import os
import oracle as orc
from bspatom_amd.namelist import read_namelists
from conftest import GOLDEN


def case_cfg(name):
    nl = read_namelists(open(os.path.join(GOLDEN, "inputs", name + ".inp")).read())
    kw = {}
    kw.update(nl["vars_bsp"]); kw.update(nl["vars_tise"])
    return orc.make_cfg(**kw)


# ---- accuracy ratchet (tests/golden/accuracy_ratchet.json) -------------------------------------------------------------
# Direct GPU-vs-truth figures per case, measured on MI355X by tools/make_ratchet.py and committed next to the truth fixtures.
# The reference-relative bars of test_gpu_solve.py::full_size_bar stay as they are; they tolerate whatever error the
# REFERENCE's LAPACK has at an eigenvalue, which next to zero is up to 1000 x what this solver achieves.  The ratchet pins
# the achieved level itself: a change of route (one- or two-step bulge chasing, another panel factorisation, approximate
# reciprocals in the reflectors) that costs accuracy shows up as a failing test, one that gains accuracy as a diff of the
# JSON when it is regenerated.
RATCHET_FILE = os.path.join(GOLDEN, "accuracy_ratchet.json")
RATCHET_FACTOR = 2.0
# floors below which a figure is not held against the solver (a stored value below the floor is rounding luck):
# 1e-10 relative is north_star's own bar; 0.02 eps lambda_max is the level LAPACK itself reaches next to zero at best
RATCHET_FLOOR_REL = 1e-10
RATCHET_FLOOR_NEAR = 0.02


def truth_stats(E_l, Eref_l, idx, tru):
    """Figures of one channel against its stored truth set: worst relative error, number of eigenvalues beyond 1e-10
    relative, and the largest absolute error among the 24 eigenvalues nearest zero in units of eps * lambda_max."""
    import numpy as np
    lam = float(np.max(np.abs(Eref_l)))
    eg = np.abs(E_l[idx] - tru)
    near = np.argsort(np.abs(tru))[:24]
    return {"worst_rel": float(np.max(eg / np.abs(tru))), "n_beyond": int(np.sum(eg > 1e-10 * np.abs(tru))),
            "near_zero": float(np.max(eg[near]) / (np.finfo(float).eps * lam)), "n_truth": int(len(idx))}


def aggregate_stats(per_channel):
    return {"worst_rel": max(s["worst_rel"] for s in per_channel), "n_beyond": sum(s["n_beyond"] for s in per_channel),
            "near_zero": max(s["near_zero"] for s in per_channel), "n_truth": sum(s["n_truth"] for s in per_channel),
            "channels": len(per_channel)}


def ratchet_check(case, per_channel, linear=True):
    """Assert the aggregated figures of `case` (list of truth_stats, one per channel, ALL its channels) against the committed
    ratchet.  Returns the message that describes both."""
    import json
    assert os.path.exists(RATCHET_FILE), "tests/golden/accuracy_ratchet.json is missing: run tools/make_ratchet.py on the GPU box"
    R = json.load(open(RATCHET_FILE))["cases"]
    assert case in R, "no ratchet entry for %s: run tools/make_ratchet.py on the GPU box" % case
    r, a = R[case], aggregate_stats(per_channel)
    assert a["channels"] == r["channels"] and a["n_truth"] == r["n_truth"], (case, a, r)
    msg = ("ratchet %s: worst rel vs truth %.2e (stored %.2e), beyond 1e-10: %d (stored %d), near-zero abs/(eps lam) %.4f (stored %.4f)"
           % (case, a["worst_rel"], r["worst_rel"], a["n_beyond"], r["n_beyond"], a["near_zero"], r["near_zero"]))
    if linear:           # grids with an exponential part: relative errors next to zero are not meaningful, near_zero is
        assert a["worst_rel"] <= max(RATCHET_FACTOR * r["worst_rel"], RATCHET_FLOOR_REL), msg
        assert a["n_beyond"] <= max(int(RATCHET_FACTOR * r["n_beyond"]), r["n_beyond"] + 2), msg
    assert a["near_zero"] <= max(RATCHET_FACTOR * r["near_zero"], RATCHET_FLOOR_NEAR), msg
    return msg
