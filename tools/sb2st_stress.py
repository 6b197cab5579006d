"""Repeat the full-size batched solve and demand bit-identical spectra: bulge chasing does the same arithmetic in
the same order whatever the timing of the paired workgroups, so any difference between runs is a race."""
import sys, numpy as np
sys.path.insert(0, "/root/repo")
sys.path.insert(0, "/root/repo/tests")
from bspatom_amd import capi
from bspatom_amd.namelist import read_namelists

def input_from_case(name, **over):
    nl = read_namelists(open(f"/root/repo/tests/golden/inputs/{name}.inp").read())
    kw = {}
    kw.update(nl["vars_bsp"]); kw.update(nl["vars_tise"]); kw.update(over)
    return capi.make_input(**kw)

def main(reps=8, nl=128):
    prob = capi.Problem(input_from_case("c4_4096", l_fin=nl - 1))
    E0, info = prob.solve(0, nl)
    bad = 0
    for r in range(1, reps):
        E, info = prob.solve(0, nl)
        diff = np.where(np.any(E != E0, axis=1))[0]
        if len(diff):
            bad += 1
            print(f"run {r}: {len(diff)} channels differ from run 0: {diff[:10]}, max |dE| {np.max(np.abs(E - E0)):.3e}", flush=True)
    print(f"{reps} runs, {bad} differing from the first; monotone lowest eigenvalue: {bool(np.all(np.diff(E0[:, 0]) > 0))}")
    prob.close()
    return bad

if __name__ == "__main__":
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    nl = int(sys.argv[2]) if len(sys.argv) > 2 else 128        # 128: pairs; 64: rings of 4; 32: rings of 4 (n = 4096)
    sys.exit(1 if main(reps, nl) else 0)
