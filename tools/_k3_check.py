import sys, os, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from bspatom_amd import capi
nl = 16
prob = capi.Problem(capi.make_input(kind_grid=0, ra=0.0, rb=800.0, k=9, nfun=4096, l_fin=nl - 1, zatom=1.0))
capi.set_option("syr2k3", 0)
E0, info = prob.solve(0, nl); assert np.all(info == 0)
capi.set_option("syr2k3", 1)
E1, info = prob.solve(0, nl); assert np.all(info == 0)
print("syr2k3 against gemm2_kernel: spectra identical:", np.array_equal(E0, E1), " max |dE| / lam", np.max(np.abs(E0 - E1)) / np.max(np.abs(E0)), flush=True)
prob.close()
prob = capi.Problem(capi.make_input(kind_grid=0, ra=0.0, rb=200.0, k=9, nfun=1000, l_fin=3, zatom=1.0))      # edge tiles (n not a multiple of 128)
capi.set_option("syr2k3", 0)
E0, info = prob.solve(0, 4); assert np.all(info == 0)
capi.set_option("syr2k3", 1)
E1, info = prob.solve(0, 4); assert np.all(info == 0)
print("n = 1000: spectra identical:", np.array_equal(E0, E1), flush=True)
