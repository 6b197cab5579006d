"""the chase kernels of the band route against each other and against scipy on random pencils: eigenvalue error of the band each
one leaves, run-to-run and batch-size differences (cw_items4 = 0: one item per wave, 1: four items per wave)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from bspatom_amd import capi
from test_gpu_stages import _random_pencil, _dense_upper, _band_eigs
import scipy.linalg as sla
for n, k, nl in [(16, 9, 1), (24, 9, 1), (32, 9, 1), (40, 9, 2), (100, 9, 2), (250, 5, 2), (1000, 9, 2)]:
    SB, HB = _random_pencil(n, k, nl, 11 * n + k)
    ref = sla.eigh(_dense_upper(HB[0]), _dense_upper(SB), eigvals_only=True)
    line = []
    for v in (0, 1):
        capi.set_option("cw_items4", v)
        a, _ = capi.stage_crawford(SB, HB)
        b, _ = capi.stage_crawford(SB, HB)
        c1, _ = capi.stage_crawford(SB, HB[:1])
        err = np.max(np.abs(_band_eigs(a[0], n, 15) - ref)) / np.max(np.abs(ref))
        line.append("v%d err %.1e rr %.0e b1 %.0e nan %d" % (v, err, np.max(np.abs(a - b)), np.max(np.abs(a[0] - c1[0])), int(np.isnan(a).sum())))
    print(n, k, nl, " | ".join(line))
