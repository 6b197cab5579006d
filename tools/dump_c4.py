import sys, numpy as np
sys.path.insert(0, ".")
from bspatom_amd import capi
p = capi.Problem(capi.make_input(kind_grid=0, rb=800.0, k=9, nfun=4096, l_fin=1))
E, info = p.solve(0, 2)
np.save("gpurun_out/E_c4_gpu.npy", E)
for v in (3, 6):
    import os
    os.environ["BSP_SB2ST_VERSION"] = str(v)
print("done", info)
