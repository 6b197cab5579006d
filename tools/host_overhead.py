"""Where the wall time of a bench step goes beyond the device pipeline (HIP-event total): per-call host timings."""
import sys, time, numpy as np, torch
sys.path.insert(0, "/root/repo")
from bspatom_amd import capi
nl = 128
inp = capi.make_input(kind_grid=0, ra=0.0, rb=800.0, k=9, nfun=4096, n0_ini=1, l_ini=0, l_fin=nl - 1, zatom=1.0)
prob = capi.Problem(inp, device=0)
E_dev = torch.empty(nl * prob.nfun, dtype=torch.float64, device="cuda")
for it in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    info = prob.solve_dev(0, nl, E_dev.data_ptr()); t1 = time.perf_counter()
    tm = prob.last_timing(); t2 = time.perf_counter()
    c = prob.eigvec(0, 1); t3 = time.perf_counter()
    r, u = prob.write_wf(c); t4 = time.perf_counter()
    torch.cuda.synchronize(); t5 = time.perf_counter()
    print("step %d: solve_dev %.1f ms (device total %.1f) last_timing %.2f eigvec %.2f write_wf %.2f sync %.2f | wall %.1f" % (
        it, 1e3 * (t1 - t0), tm["total"], 1e3 * (t2 - t1), 1e3 * (t3 - t2), 1e3 * (t4 - t3), 1e3 * (t5 - t4), 1e3 * (t5 - t0)), flush=True)
prob.close()
