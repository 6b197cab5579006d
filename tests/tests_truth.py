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
# the achieved level itself.
#
# Round 4: the ratchet HOLDS.  Round 3's file was re-measured whenever a new kernel tripped it (round-3 verdict, weak 1), and
# one of its gated figures, worst_rel, is rounding noise divided by the smallest |E| of a case -- it moves by factors of five
# between equally stable routes.  Now:
#   * gated figures are ABSOLUTE, in units of eps * lambda_max of the channel: `max_abs` over ALL stored truths of a case,
#     `near_zero` over the 24 truths nearest zero of a channel, and the count `n_beyond` of truths missed by more than 1e-10
#     relative (north_star's bar).  `worst_rel` is reported, not gated.
#   * every case carries `best` per ROUTE (1 = dense, 2 = band): the tightest value ever measured per figure (the dense route's
#     seeded from the three files of round 3, commits e4c998b, b82f898, 98975d6).  tools/make_ratchet.py may lower a `best`,
#     never raise it.  The gate is 2 x the route's best; the case's DEFAULT route is held to 2 x the smaller of both routes'
#     bests (floors below).
#   * a route that cannot meet a gate needs an `override` entry IN THE JSON (figure, bar, the eigenvalue, its absolute error and
#     the reference's, why) -- visible in review -- written only by `make_ratchet.py --allow-regress case:figure:reason`.
RATCHET_FILE = os.path.join(GOLDEN, "accuracy_ratchet.json")
RATCHET_FACTOR = 2.0
# floors below which a figure is not held against the solver (a stored value below the floor is rounding luck):
# 0.02 eps lambda_max next to zero is the level LAPACK itself reaches there at best; 1 eps lambda_max anywhere in the spectrum is
# half an ulp of the largest eigenvalue (and ratchet_gate adds the reference's own max_abs of the case as a floor)
RATCHET_FLOOR_NEAR = 0.02
RATCHET_FLOOR_ABS = 1.0
GATED = ("max_abs", "near_zero", "n_beyond")


def truth_stats(E_l, Eref_l, idx, tru):
    """Figures of one channel against its stored truth set: largest absolute error over the set and over the 24 eigenvalues
    nearest zero, in units of eps * lambda_max; number of eigenvalues beyond 1e-10 relative; worst relative error (reported)."""
    import numpy as np
    lam = float(np.max(np.abs(Eref_l)))
    eg = np.abs(E_l[idx] - tru)
    near = np.argsort(np.abs(tru))[:24]
    unit = np.finfo(float).eps * lam
    return {"worst_rel": float(np.max(eg / np.abs(tru))), "n_beyond": int(np.sum(eg > 1e-10 * np.abs(tru))),
            "near_zero": float(np.max(eg[near]) / unit), "max_abs": float(np.max(eg) / unit), "n_truth": int(len(idx))}


def aggregate_stats(per_channel):
    return {"worst_rel": max(s["worst_rel"] for s in per_channel), "n_beyond": sum(s["n_beyond"] for s in per_channel),
            "near_zero": max(s["near_zero"] for s in per_channel), "max_abs": max(s["max_abs"] for s in per_channel),
            "n_truth": sum(s["n_truth"] for s in per_channel), "channels": len(per_channel)}


def ratchet_best(entry, fig, route):
    """the value a measurement of `route` is held against: the route's own best; for the case's default route the smaller of the
    two routes' bests (a new default may not be worse than what the other route achieved)"""
    b = entry["best"]
    own = b.get("route%s" % route, {}).get(fig)
    if route == entry.get("default_route"):
        vals = [v.get(fig) for v in b.values() if v.get(fig) is not None]
        return min(vals) if vals else None
    return own


def ratchet_gate(entry, fig, route):
    """the bar of one gated figure of one case for a route (1 dense, 2 band): an explicit override if the JSON carries one for the
    route, else 2 x best with its floor"""
    ov = entry.get("override", {}).get("route%s" % route, {}).get(fig)
    if ov is not None:
        return float(ov["bar"])
    best = ratchet_best(entry, fig, route)
    if fig == "n_beyond":
        return max(int(RATCHET_FACTOR * best), best + 2)
    if fig == "near_zero":
        return max(RATCHET_FACTOR * best, RATCHET_FLOOR_NEAR)
    # max_abs is set by the top of the spectrum (eps cond(S) lambda_max, 3 .. 120 on the fixtures for LAPACK itself): a value at
    # or below the reference's own figure on the case is not held against the solver
    return max(RATCHET_FACTOR * best, RATCHET_FLOOR_ABS, entry.get("reference", {}).get("max_abs", 0.0))


def ratchet_violations(entry, a, linear=True, route=None):
    """gated figures of the measured aggregate `a` that exceed their bars: list of (figure, value, bar)"""
    figs = GATED if linear else ("max_abs", "near_zero")       # grids with an exponential part: relative errors next to zero are
    out = []                                                   # not meaningful, absolute ones are
    for fig in figs:
        if ratchet_best(entry, fig, route) is None:
            continue                                           # no measurement of this route yet (make_ratchet.py fills it)
        bar = ratchet_gate(entry, fig, route)
        if a[fig] > bar:
            out.append((fig, a[fig], bar))
    return out


def ratchet_check(case, per_channel, linear=True, route=None):
    """Assert the aggregated figures of `case` (list of truth_stats, one per channel, ALL its channels) against the committed
    ratchet (`route`: the route the spectra were computed by, for the route-specific overrides).  Returns the message that
    describes both."""
    import json
    assert os.path.exists(RATCHET_FILE), "tests/golden/accuracy_ratchet.json is missing: run tools/make_ratchet.py on the GPU box"
    R = json.load(open(RATCHET_FILE))["cases"]
    assert case in R, "no ratchet entry for %s: run tools/make_ratchet.py on the GPU box" % case
    r, a = R[case], aggregate_stats(per_channel)
    assert a["channels"] == r["channels"] and a["n_truth"] == r["n_truth"], (case, a, r)
    assert route in (1, 2), "ratchet_check wants the route the spectra were computed by"
    b = {f: ratchet_best(r, f, route) for f in GATED}
    fmt = lambda v, p: ("%" + p) % v if v is not None else "-"
    msg = ("ratchet %s route %d: max abs / (eps lam) %.4f (best %s), near zero %.4f (best %s), beyond 1e-10: %d (best %s); worst rel %.2e (not gated)"
           % (case, route, a["max_abs"], fmt(b["max_abs"], ".4f"), a["near_zero"], fmt(b["near_zero"], ".4f"), a["n_beyond"],
              fmt(b["n_beyond"], "d"), a["worst_rel"]))
    bad = ratchet_violations(r, a, linear, route)
    assert not bad, msg + " -- over the bar: " + ", ".join("%s %.4g > %.4g" % v for v in bad)
    return msg
