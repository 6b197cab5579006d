import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from bspatom_amd import capi
from test_gpu_stages import _random_pencil, _dense_upper, _band_eigs
import scipy.linalg as sla
for n, k, nl in [(16, 9, 1), (24, 9, 1), (32, 9, 1), (40, 9, 1), (40, 9, 2), (100, 9, 2)]:
    SB, HB = _random_pencil(n, k, nl, 11 * n + k)
    outs = []
    for v in (0, 1, 1, 1):
        capi.set_option("cw_items4", v)
        AB, info = capi.stage_crawford(SB, HB)
        outs.append(AB)
    AB1, _ = capi.stage_crawford(SB, HB[:1])
    ref = sla.eigh(_dense_upper(HB[0]), _dense_upper(SB), eigvals_only=True)
    errs = [np.max(np.abs(_band_eigs(o[0], n, 15) - ref)) / np.max(np.abs(ref)) for o in outs]
    d12 = np.max(np.abs(outs[1] - outs[2])); d23 = np.max(np.abs(outs[2] - outs[3])); d1 = np.max(np.abs(outs[1][0] - AB1[0]))
    w = np.argwhere(outs[1] != outs[2])
    print(n, k, nl, "eig errs (old, new x3)", ["%.1e" % e for e in errs], "run-to-run diffs %.1e %.1e, batch-1 diff %.1e" % (d12, d23, d1),
          "first differing entries", w[:4].tolist(), "nan:", [int(np.isnan(o).sum()) for o in outs])
