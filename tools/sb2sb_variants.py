import sys, numpy as np
import os; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from bspatom_amd import capi
from test_gpu_stages import _random_band64
for n, npad, batch in ((100,128,1),(448,448,2),(1000,1024,2),(2048,2048,1)):
    AB = _random_band64(n, npad, batch, n)
    capi.set_option("sb2sb_mfma", 1); a = capi.stage_sb2sb(AB, n)
    capi.set_option("sb2sb_mfma", 3); b = capi.stage_sb2sb(AB, n)
    capi.set_option("sb2sb_mfma", 1)
    print(n, "bit-identical:", np.array_equal(a, b), "max diff", np.max(np.abs(a-b)))
