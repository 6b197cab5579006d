#!/usr/bin/env python3
"""truth_report.py -- error of the GPU spectra and of the reference's LAPACK spectra against the 113-bit truth
fixtures (tests/golden/truth_*.npz), per case and channel.  Run on the GPU box:

    python tools/truth_report.py [case ...] > gpurun_out/truth_report.txt
"""
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from bspatom_amd import capi                      # noqa: E402
from bspatom_amd.namelist import read_namelists  # noqa: E402

G = os.path.join(ROOT, "tests", "golden")


def inp_of(name):
    nl = read_namelists(open(os.path.join(G, "inputs", name + ".inp")).read())
    kw = {}
    kw.update(nl["vars_bsp"]); kw.update(nl["vars_tise"])
    return capi.make_input(**kw)


def main():
    cases = sys.argv[1:] or ["c3_1024_l31", "c5_1024_k11", "lin1024", "c2_2048", "c4_4096"]
    eps = np.finfo(float).eps
    for name in cases:
        tp = os.path.join(G, "truth_" + name + ".npz")
        if not os.path.exists(tp):
            continue
        t = np.load(tp); g = np.load(os.path.join(G, name + ".npz"))
        Eref = g["E"]
        nch = Eref.shape[0]
        prob = capi.Problem(inp_of(name))
        E, info = prob.solve(0, nch)
        assert np.all(info == 0)
        print("== %s: n=%d, %d channels; timing %s" % (name, Eref.shape[1], nch, prob.last_timing()))
        for l in range(nch):
            lam = np.max(np.abs(Eref[l]))
            d = np.abs(E[l] - Eref[l]); rel = d / np.abs(Eref[l])
            exc = np.where(rel > 1e-10)[0]
            sel = t["chan"] == l
            idx = t["idx"][sel]; tru = t["hi"][sel]
            eg = np.abs(E[l][idx] - tru); er = np.abs(Eref[l][idx] - tru)
            rg = eg / np.abs(tru); rr = er / np.abs(tru)
            near = np.argsort(np.abs(tru))[:24]
            print("l=%2d vs ref: worst rel %.2e normwise %.2e exceptions(>1e-10) %d %s | vs truth: gpu worst rel %.2e (exc %d)  ref worst rel %.2e (exc %d) | "
                  "near-zero abs err/(eps lam): gpu max %.4f med %.4f  ref max %.4f med %.4f | all-truth abs/(eps lam): gpu %.2f ref %.2f"
                  % (l, rel.max(), d.max() / lam, len(exc), "in-truth-set" if set(exc) <= set(idx) else "OUTSIDE-truth-set",
                     rg.max(), int(np.sum(rg > 1e-10)), rr.max(), int(np.sum(rr > 1e-10)),
                     eg[near].max() / (eps * lam), np.median(eg[near]) / (eps * lam),
                     er[near].max() / (eps * lam), np.median(er[near]) / (eps * lam),
                     eg.max() / (eps * lam), er.max() / (eps * lam)))
            for i in exc[:6]:
                j = np.where(idx == i)[0]
                if len(j):
                    print("      exc idx %d E=%.6e  gpu-ref %.2e | gpu-truth %.2e ref-truth %.2e" % (i, Eref[l][i], d[i], eg[j[0]], er[j[0]]))
                else:
                    print("      exc idx %d E=%.6e  gpu-ref %.2e (no truth stored)" % (i, Eref[l][i], d[i]))
        prob.close()


if __name__ == "__main__":
    main()
