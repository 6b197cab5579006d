#!/usr/bin/env python3
"""make_ratchet.py -- measure the GPU-vs-truth figures of every case that has a 113-bit truth fixture and write
tests/golden/accuracy_ratchet.json (tests/tests_truth.py::ratchet_check holds the -m gpu tests to 2 x these).  Run on the
GPU box; the file lands in gpurun_out/ and is copied next to the truth fixtures by hand:

    python tools/make_ratchet.py [case ...] > gpurun_out/ratchet.log     # -> gpurun_out/accuracy_ratchet.json
"""
import glob
import json
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from bspatom_amd import capi                      # noqa: E402
from bspatom_amd.namelist import read_namelists  # noqa: E402
from tests_truth import truth_stats, aggregate_stats   # noqa: E402

G = os.path.join(ROOT, "tests", "golden")


def inp_of(name):
    nl = read_namelists(open(os.path.join(G, "inputs", name + ".inp")).read())
    kw = {}
    kw.update(nl["vars_bsp"]); kw.update(nl["vars_tise"])
    return capi.make_input(**kw)


def main():
    cases = sys.argv[1:] or sorted(os.path.basename(f)[6:-4] for f in glob.glob(os.path.join(G, "truth_*.npz")))
    out = {"source": "tools/make_ratchet.py on MI355X: default route of libbspatom, all channels of the case in one batch",
           "figures": "worst_rel = max |E_gpu - truth| / |truth| over the stored truth set; n_beyond = eigenvalues of the set beyond "
                      "1e-10 relative; near_zero = max |E_gpu - truth| over the 24 truth eigenvalues nearest zero of a channel, in "
                      "units of eps * lambda_max, maximum over channels", "cases": {}}
    old = os.path.join(ROOT, "gpurun_out", "accuracy_ratchet.json")
    for name in cases:
        t = np.load(os.path.join(G, "truth_" + name + ".npz")); g = np.load(os.path.join(G, name + ".npz"))
        Eref = g["E"]
        chans = sorted(int(l) for l in np.unique(t["chan"]))
        nch = max(chans) + 1
        prob = capi.Problem(inp_of(name))
        E, info = prob.solve(0, nch)
        assert np.all(info == 0)
        per = []
        for l in chans:
            sel = t["chan"] == l
            per.append(truth_stats(E[l], Eref[l], t["idx"][sel], t["hi"][sel]))
        a = aggregate_stats(per)
        a["kind_grid"] = int(inp_of(name).kind_grid)
        a["nfun"] = int(prob.nfun)
        out["cases"][name] = a
        print(name, a, flush=True)
        prob.close()
        json.dump(out, open(old, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    main()
