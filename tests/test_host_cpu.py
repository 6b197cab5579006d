"""CPU-side tests (no GPU): the C-ABI library loads and exports every symbol include/bspatom.h
declares, the host set-up code reproduces the reference's sizes/knots/Gauss-Legendre rule bit for
bit, the namelist reader accepts the bsp_0.inp grammar, output formatting matches the reference's
text, the product never routes through the oracle, and the N>1 sharding + gather works (gloo)."""
import os
import re
import subprocess
import sys
import numpy as np
import pytest
from conftest import load_golden, golden_input, SMALL_CASES, ROOT

from bspatom_amd import capi
from bspatom_amd.namelist import read_namelists, NamelistError
from bspatom_amd.host import fortran_g


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "bspatom.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b(bspatom_[a-z_0-9]+|bsp_dsygv_)\s*\(", hdr))
    assert len(names) >= 18
    L = capi.lib()
    missing = [n for n in sorted(names) if not hasattr(L, n)]
    assert not missing, missing
    assert set(capi.EXPORTS) == names


def test_lapack_shim_exports_dsygv():
    """libbspatom_lapack.so (csrc/lapack_shim.c): the plain LAPACK name the reference's link line resolves
    (matrices.f90:248, src/Makefile:23), forwarding to bsp_dsygv_; nothing else is exported under a LAPACK name.
    Without a GPU the call reports LAPACK's 'failed' class (info = n), never a silent zero."""
    import ctypes as C
    import subprocess
    so = os.path.join(ROOT, "bspatom_amd", "libbspatom_lapack.so")
    assert os.path.exists(so), "run make -C bspatom_amd/csrc"
    syms = subprocess.run(["nm", "-D", "--defined-only", so], capture_output=True, text=True).stdout
    defined = [l.split()[-1] for l in syms.split("\n") if l.strip()]
    assert "dsygv_" in defined and not [d for d in defined if d.endswith("_") and d != "dsygv_"]
    L = C.CDLL(so)
    n = 4
    a = np.eye(n, order="F"); b = np.eye(n, order="F"); w = np.zeros(n); work = np.zeros(16)
    it = C.c_int(1); nn = C.c_int(n); ld = C.c_int(n); lw = C.c_int(-1); info = C.c_int(7)
    L.dsygv_.restype = None
    args = lambda: (C.byref(it), C.c_char_p(b"V"), C.c_char_p(b"U"), C.byref(nn), a.ctypes.data_as(C.c_void_p), C.byref(ld),
                    b.ctypes.data_as(C.c_void_p), C.byref(ld), w.ctypes.data_as(C.c_void_p), work.ctypes.data_as(C.c_void_p),
                    C.byref(lw), C.byref(info), C.c_size_t(1), C.c_size_t(1))
    L.dsygv_(*args())
    assert info.value == 0 and work[0] == 3 * n - 1                  # workspace query needs no device
    if capi.lib().bspatom_device_count() == 0:
        lw.value = 16
        L.dsygv_(*args())
        assert info.value == n


def test_no_gpu_means_loud_failure():
    if capi.lib().bspatom_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(capi.BspAtomError) as ei:
        capi.Problem(capi.make_input(kind_grid=0, rb=50.0, k=7, nfun=64))
    assert ei.value.code == -4
    with pytest.raises(capi.BspAtomError):
        capi.stage_bisect(np.zeros((1, 4)), np.zeros((1, 3)))


def test_product_never_touches_the_oracle():
    bad = []
    for dp, _, fs in os.walk(os.path.join(ROOT, "bspatom_amd")):
        for f in fs:
            if f.endswith((".py", ".hip", ".cpp", ".h", ".f90")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                if re.search(r"^\s*(import|from)\s+oracle|liborc|bsp_oracle|oracle/", txt, flags=re.M):
                    bad.append(f)
    assert not bad, bad


@pytest.mark.parametrize("name", SMALL_CASES + ["lin1024", "wf_fatal", "c2_2048", "c4_4096"])
def test_host_setup_bit_exact(name):
    g = load_golden(name)
    nl = read_namelists(open(golden_input(name)).read())
    inp = capi.make_input(**{**nl["vars_bsp"], **nl["vars_tise"]})
    s, rt, aind, xg, wg = capi.host_setup(inp)
    nfun, k, ka, nkp, nointv, nbc1, nbc2, lmax = [int(v) for v in g["sizes"]]
    assert (s.nfun, s.k, s.ka, s.nkp, s.nointv, s.nbc1, s.nbc2, s.lmax) == (nfun, k, ka, nkp, nointv, nbc1, nbc2, lmax)
    assert np.array_equal(rt, g["rt"]) and np.array_equal(aind, g["aind"])
    assert np.array_equal(xg, g["xg"]) and np.array_equal(wg, g["wg"])
    assert s.npad % 64 == 0 and 0 <= s.npad - s.nfun < 64


def test_host_setup_rejects_bad_input():
    with pytest.raises(capi.BspAtomError):
        capi.host_setup(capi.make_input(kind_grid=0, rb=10.0, k=1, nfun=20))      # k < 2
    with pytest.raises(capi.BspAtomError):
        capi.host_setup(capi.make_input(kind_grid=0, rb=10.0, k=7, nfun=3))       # nfun < k
    with pytest.raises(capi.BspAtomError):
        capi.host_setup(capi.make_input(kind_grid=5, rb=10.0, k=7, nfun=30))
    # a box of no extent, rb < ra, not-a-number, an exponential grid shorter than its first interior knot, n0_ini = 0:
    # the reference runs into BSPLVB's STOP or NaN matrices with these; here they are argument errors up front
    for kw in (dict(kind_grid=0, rb=0.0, k=7, nfun=64), dict(kind_grid=0, ra=60.0, rb=50.0, k=7, nfun=64),
               dict(kind_grid=0, rb=float("nan"), k=7, nfun=64), dict(kind_grid=0, rb=50.0, zatom=float("inf"), k=7, nfun=64),
               dict(kind_grid=1, ra=0.0, rb=0.005, k=7, nfun=64), dict(kind_grid=0, rb=50.0, k=7, nfun=64, n0_ini=0),
               dict(kind_grid=0, rb=50.0, k=7, nfun=64, l_ini=-1),
               dict(kind_grid=2, rb=50.0, rmax=100.0, k=7, nfun=64), dict(kind_grid=0, rb=50.0, k=7, nfun=64, kind_pot=7)):
        with pytest.raises(capi.BspAtomError) as ei:
            capi.host_setup(capi.make_input(**kw))
        assert ei.value.code == -2, kw
    # sizes the reference accepts and this build does not (include/bspatom.h: BSPATOM_MAX_K, BSPATOM_MAX_KA): UNSUPPORTED (-5),
    # not an argument error -- the input is valid, the limit is the build's
    for kw in (dict(kind_grid=0, rb=50.0, k=7, nfun=64, ka=100), dict(kind_grid=0, rb=50.0, k=17, nfun=64)):
        with pytest.raises(capi.BspAtomError) as ei:
            capi.host_setup(capi.make_input(**kw))
        assert ei.value.code == -5, kw


def test_namelist_reference_input():
    nl = read_namelists(open(golden_input("bsp0")).read())
    assert nl["vars_bsp"] == dict(kind_grid=2, rmax=60.0, ra=0.0, rb=500.0, k=7, nfun=100, kind_bc1=0, kind_bc2=0)
    assert nl["vars_tise"]["l_fin"] == 2 and nl["vars_tise"]["emax_fin"] == 1.5 and nl["vars_tise"]["zatom"] == 1.0
    assert nl["vars_field"]["kind_pi"] == 0 and nl["vars_field"]["i0"] == 1.0e15 and nl["vars_field"]["nepts"] == -200


def test_namelist_grammar():
    txt = "junk line\n&vars_bsp KIND_GRID=1, ra = 0.5d0,\n rb=2.0E1 k=5 nfun=30 /\n! c\n&VARS_TISE l_fin=3 &END\n&VARS_FIELD /\n"
    nl = read_namelists(txt)
    assert nl["vars_bsp"] == dict(kind_grid=1, ra=0.5, rb=20.0, k=5, nfun=30)
    assert nl["vars_tise"] == dict(l_fin=3) and nl["vars_field"] == {}
    with pytest.raises(NamelistError, match="unknown key"):
        read_namelists("&VARS_BSP foo=1 &end &VARS_TISE &end &VARS_FIELD &end")
    with pytest.raises(NamelistError, match="not found"):
        read_namelists("&VARS_TISE &end &VARS_BSP k=3 &end &VARS_FIELD &end")   # wrong order
    with pytest.raises(NamelistError):
        read_namelists("&VARS_BSP k=abc &end &VARS_TISE &end &VARS_FIELD &end")


def test_g_format_matches_reference_text():
    """The eigenvalue lines the reference printed (FORMAT(T2,I4,T8,G22.15)) are reproduced verbatim."""
    g = load_golden("bsp0")
    lines = [l for l in str(g["stdout"]).split("\n") if re.match(r"^\s+\d+\s+-?\d?\.\d", l)]
    assert len(lines) == 60
    E = g["E"]
    k = 0
    for l in range(3):
        for i in range(20):
            mine = " %4d  %s" % (i + 1 + l, fortran_g(E[l, i], 22, 15))
            assert mine.rstrip() == lines[k].rstrip(), (mine, lines[k])
            k += 1
    assert fortran_g(0.05, 20, 10) == "    0.5000000000E-01"
    assert fortran_g(0.0, 20, 10) == "     0.000000000    "


# ---- N > 1: sharding and gather over gloo, world_size 2 ---------------------------------------
_WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch, torch.distributed as dist
from bspatom_amd.parallel import channel_range, gather_spectra
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%(port)d", rank=rank, world_size=world)
g = np.load(%(gold)r)
E = g["E"]; lmax = E.shape[0] - 1; nfun = E.shape[1]
for per in (None, 2):
    l0, nl = channel_range(rank, world, lmax, per)
    counts = [channel_range(r, world, lmax, per)[1] for r in range(world)]
    # stand-in for the GPU solve of this rank's channels: the reference spectra of those channels
    mine = torch.from_numpy(E[l0:l0 + nl].copy())
    allE = gather_spectra(mine, nfun, counts)
    tot = sum(counts)
    assert allE.shape == (tot, nfun), allE.shape
    assert np.array_equal(allE.numpy(), E[:tot]), "gather mismatch"
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok")
'''


@pytest.mark.parametrize("case", ["simfues", "lin256"])      # 5 channels (ragged 3+2) and 4 channels
def test_l_sharding_and_gather_gloo(tmp_path, case):
    port = 29500 + os.getpid() % 2000
    script = tmp_path / "worker.py"
    script.write_text(_WORKER % dict(root=ROOT, port=port, gold=os.path.join(ROOT, "tests", "golden", case + ".npz")))
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=180)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs


_SHARDED_WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch, torch.distributed as dist
from bspatom_amd import host
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%(port)d", rank=rank, world_size=world)
g = np.load(%(gold)r)
E = g["E"]; text = str(g["namelist"])
npts = 200
rr = np.linspace(0.0, 50.0, npts + 1); uu = np.sin(rr) * np.exp(-0.1 * rr)
l_ini = int(%(l_ini)d)
def solver(l0, nl):            # stand-in for the GPU solve of this rank's channels: the reference's spectra
    return E[l0:l0 + nl].copy(), ((rr, uu) if l0 <= l_ini < l0 + nl else None)
out = host.run_sharded(text, outdir=%(out)r, npts=npts, solver=solver)
assert (out is None) == (rank != 0)
if rank == 0:
    assert np.array_equal(out[0], E)
    open(os.path.join(%(out)r, "stdout.txt"), "w").write(out[1])
dist.barrier(); dist.destroy_process_group()
'''


@pytest.mark.parametrize("case,l_ini", [("simfues", 0), ("c1_lin", 1)])      # 5 channels over 2 ranks (3 + 2); owner of l_ini = rank 0
def test_sharded_host_gloo(tmp_path, case, l_ini):
    """host.run_sharded (the multi-GPU KIND_PI = 0 host) with world size 2 over gloo: channel blocks from
    parallel.channel_range, spectra through parallel.gather_spectra, the owner's wave-function table broadcast, rank 0
    writes -- the files must be what the single-process writer produces from the same spectra."""
    from bspatom_amd import host
    port = 29700 + os.getpid() % 2000
    script = tmp_path / "worker.py"
    outd = tmp_path / "out"; outd.mkdir()
    gold = os.path.join(ROOT, "tests", "golden", case + ".npz")
    script.write_text(_SHARDED_WORKER % dict(root=ROOT, port=port, gold=gold, out=str(outd), l_ini=l_ini))
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=180)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    g = np.load(gold)
    E = g["E"]; nch, nfun = E.shape
    ref = tmp_path / "ref"; ref.mkdir()
    rr = np.linspace(0.0, 50.0, 201); uu = np.sin(rr) * np.exp(-0.1 * rr)
    text = host.write_structure_outputs(nfun, nch - 1, E, int(l_ini), (rr, uu), str(ref))
    assert open(outd / "Enl.dat").read() == open(ref / "Enl.dat").read()
    assert open(outd / "wf_n0.dat").read() == open(ref / "wf_n0.dat").read()
    assert open(outd / "stdout.txt").read() == text
    assert os.path.isdir(outd / "CSs")


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_bench_starts_its_own_ranks_gloo(scaling):
    """`python bench.py --gpus 2` WITHOUT a launcher (the command the driver's N = 1 line generalises to): the parent must
    start `torch.distributed.run --nproc-per-node 2` as a child, stay off the GPU itself, and relay exactly one JSON line.
    Rehearsed on the CPU with --selftest-launcher (gloo, stand-in spectra, no solve): sharding, the all-gather, the barrier
    bracket, the max over ranks and the one-line output are the code the GPU run uses.  At N > 1 the line carries BOTH
    figures: the weak one (channels per GPU fixed) and configs[3] as stated (channels in total fixed); `value` is the one
    --scaling names."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--nfun", "48",
                        "--channels", "6", "--scaling", scaling, "--selftest-launcher"], cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    lines = [l for l in p.stdout.split("\n") if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), p.stdout           # stdout is the one line, nothing else
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == scaling and d["value"] > 0
    assert "no solve" in d["data"]                                           # a self-test line can never pass for a measurement
    cfg = d["config"]
    assert cfg["launched_by"] == "torch.distributed.run" and "gloo" in cfg["parallelism"]
    assert cfg["collective_calls"] == 2 * (2 + 1)                            # both shardings x (steps + warmup)
    if scaling == "weak":
        assert cfg["channels_per_gpu"] == [6, 6] and cfg["channels_total"] == 12 and "weak" in cfg["workload"]
        other = d["configs3_as_stated"]
        assert other["scaling"] == "strong" and other["channels_per_gpu"] == [3, 3] and other["channels_total"] == 6
    else:
        assert cfg["channels_per_gpu"] == [3, 3] and cfg["channels_total"] == 6 and "strong" in cfg["workload"]
        other = d["weak_scaling"]
        assert other["scaling"] == "weak" and other["channels_per_gpu"] == [6, 6] and other["channels_total"] == 12
    assert other["value"] > 0 and other["steps"] == 2


def test_bench_refuses_a_world_size_it_was_not_asked_for():
    """--gpus N under a launcher with WORLD_SIZE != N is an error, not a silent one-rank run labelled n_gpus = N."""
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--selftest-launcher"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "WORLD_SIZE" in p.stderr and not p.stdout.strip()


def test_gather_runs_the_collective_at_world_size_one():
    """With a process group the all-gather is issued at EVERY world size, 1 included (that is how the RCCL branch is
    exercised on the one-GPU box, tests/test_a_bench_rccl.py); without one nothing is exchanged."""
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import torch, torch.distributed as dist\n"
            "from bspatom_amd import parallel\n"
            "E = torch.arange(12, dtype=torch.float64)\n"
            "a = parallel.gather_spectra(E, 4, [3]); assert parallel.COLLECTIVE_CALLS == 0 and a.shape == (3, 4)\n"
            "dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%d', rank=0, world_size=1)\n"
            "b = parallel.gather_spectra(E, 4, [3]); assert parallel.COLLECTIVE_CALLS == 1 and torch.equal(a, b)\n"
            "dist.destroy_process_group()\n" % (ROOT, 29300 + os.getpid() % 200))
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stdout + p.stderr


def test_channel_range_partitions():
    from bspatom_amd.parallel import channel_range
    for world in (1, 2, 3, 8):
        for lmax in (0, 4, 127):
            seen = []
            for r in range(world):
                l0, nl = channel_range(r, world, lmax)
                seen += list(range(l0, l0 + nl))
            assert seen == list(range(lmax + 1))
    assert channel_range(3, 8, 0, per_rank=128) == (384, 128)


def test_eigenvec_all_format_cpu(tmp_path):
    """Eigenvec_All.dat writer/reader pair (matrices.f90:366-378 / ReadInputs.f90:792-830) without a GPU: records
    are I5 followed by nfun G20.10 fields; values survive to 10 significant digits."""
    from bspatom_amd import host

    class FakeProblem:
        nfun = 7
        def eigvecs(self, l, n0, count):
            rng = np.random.default_rng(l)
            return rng.standard_normal((count, self.nfun)) * 10.0 ** rng.integers(-12, 3, size=(count, 1))

    p = tmp_path / "Eigenvec_All.dat"
    host.write_eigenvec_all(str(p), FakeProblem(), 2, 5)
    lines = open(p).read().splitlines()
    assert lines[0].split() == ["7", "5", "2"]
    assert len(lines) == 1 + 3 * (1 + 5)
    assert all(len(x) == 5 + 20 * 7 for x in lines[2:7])
    nfun, n1, lmax, c = host.read_eigenvec_all(str(p))
    for l in range(3):
        Z = FakeProblem().eigvecs(l, 1, 5)
        assert np.max(np.abs(c[l] - Z) / np.abs(Z)) < 1e-9


PI3_CASES = ["pi3_emax1", "pi3_default", "pi5_emax05", "pi8_emax1", "pi3_nobound"]


@pytest.mark.parametrize("name", PI3_CASES)
def test_select_states_vs_reference(name):
    """State limits of SOLVE_SYSTEM's KIND_PI >= 3 branch (matrices.f90:290-341, :355-358) on the reference's own
    spectra (binary dump of Enl): n01, n1_max, the modified Emax_fin and the density-of-states factors rEki must be
    IDENTICAL to what the compiled reference left in its module (integers equal, doubles bit-equal); the stdout
    lines it printed must be reproduced character by character.  pi3_nobound has channels without bound states,
    where the reference carries n0_fin over from the previous channel."""
    from bspatom_amd import host
    g = load_golden(name)
    nfun, lmax, n1_max, kind_pi = (int(v) for v in g["sizes"])
    em = read_namelists(str(g["namelist"]))["vars_tise"].get("emax_fin", -1.0)
    lim = host.select_states(g["E"], em, kind_pi)
    assert np.array_equal(lim.n01, g["n01"])
    assert lim.n1_max == n1_max
    assert lim.emax_fin == g["emax_fin_out"][0]
    assert np.array_equal(lim.reki, g["reki"])
    mine = []
    for l in range(lmax + 1):
        mine.append("NUMBER OF BOUND STATES:%3d" % lim.nbold[l])
        mine.append("LIMITS FOR l =%3d STATE%5d%5d" % (l, lim.n01[l, 0] + l, lim.n01[l, 1] + l))
    mine.append("n1_max =%5d" % lim.n1_max)
    assert mine == [x.rstrip() for x in str(g["limits"]).split("\n")]


@pytest.mark.parametrize("name", PI3_CASES)
def test_eigenvec_all_text_matches_reference_file(tmp_path, name):
    """The writer on the reference's own eigenvector values must reproduce the reference's Eigenvec_All.dat text:
    header and channel lines (list-directed, flang spelling) and the first record (I5 + nfun G20.10 fields)."""
    from bspatom_amd import host
    g = load_golden(name)
    nfun, lmax, n1_max, _ = (int(v) for v in g["sizes"])
    C = g["C"]

    class RefVectors:
        pass
    rv = RefVectors(); rv.nfun = nfun
    rv.eigvecs = lambda l, n0, count: C[l, n0 - 1: n0 - 1 + count]
    p = tmp_path / "Eigenvec_All.dat"
    host.write_eigenvec_all(str(p), rv, lmax, n1_max)
    lines = open(p).read().split("\n")
    assert "\n".join(lines[:2]) == str(g["eva_head"])
    assert lines[2] == str(g["eva_row"])
    assert len(lines) == 1 + (lmax + 1) * (1 + n1_max) + 1
    _, _, _, c = host.read_eigenvec_all(str(p))
    assert np.array_equal(c, C)                      # 10-digit text -> the same doubles the fixture parsed


def test_enl_reader_on_reference_text(tmp_path):
    """Enl.dat as the reference wrote it (rebuilt from the golden stdout/E values in its FORMAT(T2,I4,T8,G22.15))
    through the READ_FR-style reader: 15 significant digits come back, limits follow ReadInputs.f90:305-312."""
    from bspatom_amd import host
    g = load_golden("pi3_emax1")
    E = g["E"]; nl, nfun = E.shape
    p = tmp_path / "Enl.dat"
    with open(p, "w") as f:
        f.write(" %d\n" % nfun)
        for l in range(nl):
            for i in range(nfun):
                f.write(" %4d  %s\n" % (i + 1, fortran_g(E[l, i], 22, 15)))
    nf, Er, n01 = host.read_enl(str(p), nl - 1, emax_fin=1.0)
    assert nf == nfun
    assert np.max(np.abs(Er - E) / np.abs(E)) < 1e-14
    # the reader's limits differ from SOLVE_SYSTEM's by the +1 the latter adds (matrices.f90:315-316)
    assert np.array_equal(n01[:, 1], g["n01"][:, 1] - 1)
    assert np.array_equal(n01[:, 2], g["n01"][:, 2])


@pytest.mark.parametrize("name", ["ta_len_s", "ta_vel_s", "ta_len_p", "ta_vel_p"])
def test_trans_amp_host_logic_vs_reference(name):
    """Host arithmetic of the KIND_PI = 1, 2 branch on the reference's spectra: SEL_LM's final channel (grid.f90:128-143),
    the final-state window (matrices.f90:272-283), the 3j symbols against the oracle's THREE_J restatement, and the
    reference's stdout lines character by character."""
    from bspatom_amd import host
    import oracle as orc
    g = load_golden(name)
    nfun, kp, n0i, l0, m0, lf, mf, mph, n0f, n1f = (int(v) for v in g["head"])
    assert host.final_channels(kp, l0, m0)[-1] == (lf, mf)
    em = read_namelists(str(g["namelist"]))["vars_tise"].get("emax_fin", -1.0)
    assert host.final_state_limits(g["E_fin"], em)[:2] == (n0f, n1f)
    for args in [(lf, 1, l0, -mf, mph, m0), (lf, 1, l0, 0, 0, 0), (lf, 1, l0, 1, -1, 0), (3, 1, 2, -2, 1, 1)]:
        assert abs(host.three_j(*args) - orc.three_j(*args)) <= 1e-15
    ref = [x.strip() for x in str(g["lines"]).split("\n")]
    assert "LIMITS FOR FINAL STATE (l=%2d) : %4d%4d" % (lf, n0f, n1f) in ref
    assert "Initial State:%3d%3d%3d" % (n0i + l0, l0, m0) in ref
    assert "Calculating Transition Amplitudes" in ref


def test_committed_bench_line_keeps_the_contract():
    """profiles/r02_bench_line.json is the line bench.py printed on the MI355X: the keys the driver and the judge read,
    the metric of BASELINE.json, a roofline object whose numbers are consistent with each other (the whole path against the
    fp64 peak, SURVEY 8d, plus a per-kernel list that names its sources), a CPU baseline measured at the metric's nfun."""
    import json
    d = json.load(open(os.path.join(ROOT, "profiles", "r02_bench_line.json")))
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert base["metric"].startswith(d["metric"])
    assert d["dtype"] == "f64" and d["scaling"] == "weak" and d["vs_baseline"] is None and d["higher_is_better"] is True
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 78.6 * d["n_gpus"]
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    n, k = 4096, 9
    F = 4.0 / 3.0 * n ** 3 + 4.0 * n ** 2 * k                        # SURVEY 8(d) flop per l-channel
    assert abs(r["achieved"] - F * d["value"] / 1e12) < 1e-9 * r["achieved"]
    names = [kk["kernel"] for kk in r["kernels"]]
    assert names[0].startswith("sb2sb") and any(x.startswith("gemm2_kernel<128") for x in names) and any(x.startswith("gemm2_kernel<64") for x in names)
    for kk in r["kernels"]:
        assert abs(kk["frac"] - kk["achieved"] / kk["peak"]) < 1e-12
        assert kk["traffic"] is None or (kk["traffic"] > 0 and kk["traffic_source"].startswith("profiles/r02_"))
        assert kk["launch_ms_source"]
    sb = r["kernels"][0]                                              # the bulge chasing in two steps: model bytes and PMC bytes agree
    assert sb["bound"] == "hbm" and sb["bytes_min"] < sb["bytes_model"] == sb["bytes_model_sb2sb"] + sb["bytes_model_sb16st"]
    assert 0.8 * sb["bytes_model"] <= sb["traffic"] <= 1.2 * sb["bytes_model"]
    sp = sb["split_from_profile"]
    assert sp["sb16st_kernel"]["launches_per_step"] == 1 and sp["sb2sb_mfma_kernel"]["launches_per_step"] > 700
    assert abs(sp["sb2sb_mfma_kernel"]["kernel_ms_per_step"] + sp["sb16st_kernel"]["kernel_ms_per_step"] - sb["launch_ms"]) < 0.05 * sb["launch_ms"]
    assert r["traffic"] > sb["traffic"]                              # the whole step's HBM bytes
    # value = channels per step / time per step
    assert abs(d["value"] - d["config"]["channels_total"] / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and "nfun=4096" in c["sample"]
    assert "value_scaled_to_nfun4096" not in c                       # measured at the metric's size, not extrapolated


@pytest.mark.parametrize("kind_pi", [3, 5, 8])
def test_handoff_files_against_the_reference_reader(tmp_path, kind_pi):
    """SURVEY 8(f).3, pinned from the CONSUMER's side (tests/golden/handoff.npz, make_golden.py --handoff): Enl.dat and
    CSs/MatElem_All.dat as this repo's writers produce them were read by the reference's own READ_COUP (ReadInputs.f90:277-369,
    unmodified, oracle/ref/ref_handoff_driver.f90) for nfields = 1, 2, 5 (KIND_PI = 3, 5, 8).  Here: (a) today's writers still
    produce exactly the texts that reader consumed; (b) what it read equals what was written, to the files' digits (G22.15,
    G20.10), including the (1, n1_fin, n0_fin) limits it rebuilds from Enl.dat and which components it keeps (the first one; all
    five only for KIND_PI >= 8, :351-358); (c) this repo's readers return the same; (d) MatElem_All.dat written with the
    reference's two WRITE forms (PhotoIon.f90:255-266: list-directed header, FORMAT(2I8,X,20G20.10)) by the Fortran runtime is
    byte-identical to the Python writer's; (e) the one MatElem_All.dat the reference's own TRANS_AMP can write in the build
    container (KIND_PI = 3 without the angular integrals, which abort there: header only) is read by read_matelem_all."""
    from bspatom_amd import host
    g = load_golden("handoff")
    t = "pi%d_" % kind_pi
    E, zT, emax, n1_max, nf = g[t + "E"], g[t + "zT"], float(g[t + "emax_fin"]), int(g[t + "n1_max"]), int(g[t + "nfields"])
    lmax, nfun = E.shape[0] - 1, E.shape[1]
    rr = np.linspace(0.0, 1.0, 3)
    host.write_structure_outputs(nfun, lmax, E, 0, (rr, rr), str(tmp_path))
    me = tmp_path / "CSs" / "MatElem_All.dat"
    host.write_matelem_all(str(me), n1_max, zT)
    assert open(tmp_path / "Enl.dat").read() == str(g[t + "enl_text"])                          # (a)
    assert open(me).read() == str(g[t + "matelem_text"])
    assert str(g[t + "matelem_text_fortran"]) == str(g[t + "matelem_text"])                    # (d)
    rnfun, rn1, rnbra, rnket, rnf, rlmax = (int(x) for x in g[t + "read_header"])             # (b)
    assert (rnfun, rn1, rnbra, rnket, rnf, rlmax) == (nfun, n1_max, zT.shape[0], zT.shape[1], nf, lmax)
    assert np.max(np.abs(g[t + "read_E"] - E) / np.abs(E)) < 1e-14
    rz = g[t + "read_z"]
    keep = nf if kind_pi >= 8 else 1
    iu = np.triu_indices(zT.shape[0])
    for c in range(nf):
        want = zT[:, :, c][iu] if c < keep else np.zeros(len(iu[0]))
        scale = np.maximum(np.abs(zT[:, :, c][iu]), 1e-300)
        assert np.max(np.abs(rz[:, :, c][iu] - want) / scale) < 1e-9
    assert np.all(rz[np.tril_indices(zT.shape[0], -1)] == 0)                                    # records cover jket >= ibra only
    n_, E2, n01 = host.read_enl(str(tmp_path / "Enl.dat"), lmax, emax)                         # (c)
    assert n_ == nfun and np.array_equal(E2, g[t + "read_E"]) and np.array_equal(n01, g[t + "read_n01"])
    n1r, z2 = host.read_matelem_all(str(me), nfields=nf)
    assert n1r == n1_max
    for c in range(keep):
        assert np.array_equal(z2[:, :, c][iu], rz[:, :, c][iu])
    hdr = tmp_path / "ref_header_only.dat"                                                      # (e)
    hdr.write_text(str(g["ref_pi3_matelem_text"]))
    assert str(g["ref_pi3_matelem_text"]) == " 44 396 0\n"
    n1h, zh = host.read_matelem_all(str(hdr), nfields=1)
    assert n1h == 44 and zh.shape == (396, 0, 1)


def test_matelem_all_round_trip(tmp_path):
    """CSs/MatElem_All.dat (PhotoIon.f90:255-266 writer, ReadInputs.f90:324-366 reader): header `n1_max nbra nket`, records
    FORMAT(2I8,X,20G20.10) for jket >= ibra; READ_COUP reads them list-directed, 2 nfields reals per record."""
    from bspatom_amd import host
    rng = np.random.default_rng(5)
    for ncomp in (1, 2, 5):
        z = rng.standard_normal((6, 6, ncomp)) + 1j * rng.standard_normal((6, 6, ncomp))
        z[2, 3, 0] = 0.0
        z[1, 4, 0] = 1.2345678901e-7 - 9.87e12j
        path = tmp_path / ("MatElem_All_%d.dat" % ncomp)
        host.write_matelem_all(str(path), 3, z)
        lines = open(path).read().split("\n")
        assert lines[0].split() == ["3", "6", "6"]
        assert len(lines[1]) == 17 + 40 * ncomp and lines[1][:17] == "       1       1 "
        assert len([l for l in lines[1:] if l]) == 6 * 7 // 2
        n1, back = host.read_matelem_all(str(path), nfields=ncomp)
        assert n1 == 3
        iu = np.triu_indices(6)
        assert np.allclose(back[iu], z[iu], rtol=1e-9, atol=0)
        assert not np.any(back[np.tril_indices(6, -1)])


def test_fortran_g_with_exponent_width():
    """G20.10E3 (CROSS_SECTIONS, FORMAT 400): F editing leaves e + 2 = 5 blanks, E editing has three exponent digits."""
    from bspatom_amd.host import fortran_g
    assert fortran_g(1.0, 20, 10, 3) == "    1.000000000     "
    assert fortran_g(1.5e-5, 20, 10, 3) == "   0.1500000000E-004"
    assert fortran_g(-2.5e12, 20, 10, 3) == "  -0.2500000000E+013"
    assert fortran_g(0.0, 20, 10, 3) == "    0.000000000     "
    assert fortran_g(0.25, 20, 10) == "    0.2500000000    "


def test_two_step_band_reduction_prototype():
    """tools/proto_sbr.py is the dense NumPy statement of what csrc/sbr2.hip does (block bulge chasing 64 -> 16 in wavefront order
    with a lag of 3 between sweeps, then the one-column chase 16 -> 1): the wavefront order never runs two items with overlapping
    tiles, the working band stays inside the 128 rows per column of the band array, the result has half-width 16 resp. 1, the
    eigenvalues survive -- and a lag of 2 does produce overlapping tiles (the kernels' LAG = 3 is not arbitrary)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("proto_sbr", os.path.join(ROOT, "tools", "proto_sbr.py"))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    import numpy as np
    n = 320
    rng = np.random.default_rng(3)
    A = np.zeros((n, n))
    for d in range(m.B + 1):
        v = rng.standard_normal(n - d)
        A += np.diag(v, -d) + (np.diag(v, d) if d else 0)
    ev0 = np.linalg.eigvalsh(A)
    st = m.sb2sb(A)
    assert st["max_bw"] <= 2 * m.B - 1 and m.bandwidth(A, 1e-13 * np.max(np.abs(ev0))) == m.D
    assert np.max(np.abs(np.linalg.eigvalsh(A) - ev0)) <= 1e-13 * np.max(np.abs(ev0))
    A16 = np.triu(np.tril(A, m.D), -m.D)
    m.chase_to_tridiagonal(A16, m.D)
    T = np.triu(np.tril(A16, 1), -1)
    assert m.bandwidth(A16, 1e-13 * np.max(np.abs(ev0))) == 1
    assert np.max(np.abs(np.linalg.eigvalsh(T) - ev0)) <= 1e-13 * np.max(np.abs(ev0))
    m.LAG = 2
    try:
        with pytest.raises(AssertionError):
            m.sb2sb(np.array(A16 + 0.0) * 0 + np.triu(np.tril(rng.standard_normal((n, n)), m.B), -m.B))
    finally:
        m.LAG = 3


def test_no_scratch_in_the_kernels_of_the_default_route():
    """Round-3 verdict, item 6: the code-object notes of libbspatom.so (llvm-readelf --notes through tools/codeobj_notes.py) -- a
    kernel that the default route of a BASELINE config launches must not use scratch (private_segment_fixed_size = 0).  The band
    route (configs 1-4: assembly, crawford.hip, the one-column chase on tiles of 8, bisection, the consumed eigenvector, WRITE_WF) has none.  The
    dense route (config 5, k = 11) still has the listed, capped exceptions; nothing else in the library may spill, and the
    listed ones may not grow."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import codeobj_notes
    ks = codeobj_notes.kernels(os.path.join(ROOT, "bspatom_amd", "libbspatom.so"))
    assert len(ks) > 60
    # kernel (substring of the demangled name) -> bytes of scratch it may have at most, and why it is tolerated
    allowed = {
        "tsqr_tree_kernel<2>": 176,     # dense route, panels <= 2048 rows: 43 registers spilled; scratch stores before the level loop,
        "tsqr_apply_kernel<2>": 208,    # one dword reloaded per column step; the uncapped <1> instances have none
        "panel_qr_kernel<16>": 508,     # dense route, panels of 4097 .. 8192 rows (n > 4160 only: config 5)
        "panel_qr_kernel<8>": 76,       # BSP_PANEL_QR=1 cross-check
        "panel_qr2_kernel<16>": 436,    # instantiated, never launched (launch_pq: RPT <= 8)
        "sb2st_kernel_v7<1>": 40,       # the instrumented instance of BSP_SB2ST_DIAG
    }
    band_route = ["point_table_kernel", "band_kernel", "crawford_item_kernel", "crawford_setup_kernel", "crawford_init_kernel",
                  "crawford_band_kernel", "crawford_corner_kernel", "crawford_flip_kernel", "band_cholesky_kernel", "sbr_rows_kernel<8, false>",
                  "sbr_rows_kernel<16, false>", "band_tail_zero_kernel",
                  "bisect3_kernel", "bisect_one3_kernel", "invit_kernel", "wf_kernel"]
    seen = {b: 0 for b in band_route}
    bad = []
    for name, v in ks.items():
        cap = next((c for a, c in allowed.items() if a in name), 0)
        if (v["private_segment_fixed_size"] or 0) > cap:
            bad.append((name, v["private_segment_fixed_size"], cap))
        for b in band_route:
            if b in name:
                seen[b] += 1
                assert not v["private_segment_fixed_size"] and not v["vgpr_spill_count"], (name, v)
    assert not bad, bad
    assert all(seen.values()), seen


def test_accuracy_ratchet_file_only_tightens():
    """tests/golden/accuracy_ratchet.json (round-3 verdict, item 4): every case carries `best` for the gated figures, no `best` is
    looser than the three files of round 3 it was seeded from, and an override names its route, its bar and a reason."""
    import json
    from tests_truth import GATED
    doc = json.load(open(os.path.join(ROOT, "tests", "golden", "accuracy_ratchet.json")))
    seed = {"c4_4096_l127": (0.08173091737950636, 33), "sf2048": (0.019458441628511368, 3), "c3_2048_l31": (0.08051401133617608, 9),
            "c3_1024_l31": (0.0741926373941524, 3), "c2_2048": (0.08051401133617608, 0), "c5_8192": (0.057375, 2)}
    assert len(doc["cases"]) >= 22
    for name, e in doc["cases"].items():
        assert e["default_route"] in (1, 2) and "route%d" % e["default_route"] in e["best"], name
        for b in e["best"].values():
            assert set(GATED) <= set(b), name
        if name in seed:                                                        # the dense route's history
            b = e["best"]["route1"]
            assert b["near_zero"] <= seed[name][0] * (1 + 1e-12) + (1e-3 if name == "c5_8192" else 0), (name, b)
            assert b["n_beyond"] <= seed[name][1], (name, b)
        for route, figs in e.get("override", {}).items():
            assert route in ("route1", "route2"), (name, route)
            for fig, o in figs.items():
                assert fig in GATED and o["bar"] >= o["measured"] and len(o["why"]) > 40, (name, route, fig)


def test_run_token_names_the_launch():
    """bspatom_run_token (csrc/comm.hip): the id that names the exchange files and the ncclUniqueId file of one launch -- "<pid of
    the launcher>.<its start time>": the same in every child of one parent, the parent's pid in front, another value for a child
    of another parent.  (No GPU needed: the call touches /proc only.)"""
    import subprocess, sys
    code = ("import ctypes, os, sys; sys.path.insert(0, %r); from bspatom_amd import capi; b = ctypes.create_string_buffer(128); "
            "assert capi.lib().bspatom_run_token(b, 128) == 0; print(os.getppid(), b.value.decode())" % ROOT)
    outs = [subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120) for _ in range(2)]
    for o in outs:
        assert o.returncode == 0, o.stderr
    (pp1, t1), (pp2, t2) = [o.stdout.split() for o in outs]
    assert pp1 == pp2 == str(os.getpid()) and t1 == t2 and t1.split(".")[0] == pp1 and int(t1.split(".")[1]) > 0
    # a child of another parent (a shell in between) gets another token
    o3 = subprocess.run(["sh", "-c", "%s -c %s; true" % (sys.executable, "'" + code.replace("'", '"') + "'")], capture_output=True, text=True, timeout=120)
    assert o3.returncode == 0 and o3.stdout.split()[1] != t1, o3.stdout + o3.stderr
    b = __import__("ctypes").create_string_buffer(4)
    assert capi.lib().bspatom_run_token(b, 4) != 0                     # a buffer that cannot hold it is an argument error


def test_band_reduction_leaves_half_width_b():
    """What csrc/crawford.hip's hand-over of a band of half-width 8 rests on, on the dense statement of the blocked reduction
    (tools/proto_crawford.py::crawford_block): the block tridiagonal result has UPPER TRIANGULAR sub-diagonal blocks E_1 .. E_{N-2}
    (half-width b) and one full block, E_0 -- also with a ragged last block -- and the eigenvalues are the pencil's."""
    import scipy.linalg as sla
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import proto_crawford as pc
    rng = np.random.default_rng(4)
    for n, b in [(64, 8), (67, 8), (41, 4), (30, 3)]:
        def band(shift):
            M = np.zeros((n, n))
            for d in range(b + 1):
                v = rng.standard_normal(n - d)
                i = np.arange(n - d)
                M[i, i + d] = v
                M[i + d, i] = v
            return M + shift * np.eye(n)
        S, H = band(2.0 * b + 4.0), band(0.0)
        A = pc.crawford_block(S, H, b)
        i, j = np.indices(A.shape)
        scale = np.max(np.abs(A))
        assert np.max(np.abs(A[np.abs(i - j) > 2 * b - 1])) == 0.0
        beyond_b = (np.abs(i - j) > b)
        assert np.max(np.abs(A[beyond_b & (np.minimum(i, j) >= b)])) <= 1e-14 * scale        # E_1 ..: triangular to rounding
        assert np.max(np.abs(A[beyond_b & (np.minimum(i, j) < b)])) > 1e-3 * scale           # E_0: full
        # the 8 x 8 RQ at the end: E_0 = R Q^T, D_0 <- Q^T D_0 Q
        R, Q = sla.rq(A[b:2 * b, :b])
        T = np.eye(n)
        T[:b, :b] = Q.T
        A2 = T.T @ A @ T
        A2[beyond_b] = 0.0
        ref = sla.eigh(H, S, eigvals_only=True)
        assert np.max(np.abs(np.linalg.eigvalsh(A2) - ref)) <= 1e-13 * np.max(np.abs(ref))


def test_band_reduction_from_both_ends():
    """The dense statement of csrc/crawford.hip's run from both ends (tools/proto_crawford_split.py::crawford_split; its own
    assertions check that what is left of S after the two independent parts is the identity plus ONE block next to the cut): half
    the chase items of the one-sided process, the pencil's eigenvalues, half-width b away from the two end blocks."""
    import scipy.linalg as sla
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import proto_crawford_split as ps
    rng = np.random.default_rng(9)
    for n, b in [(64, 8), (96, 8), (48, 4)]:
        def band(shift):
            M = np.zeros((n, n))
            for d in range(b + 1):
                v = rng.standard_normal(n - d)
                i = np.arange(n - d)
                M[i, i + d] = v
                M[i + d, i] = v
            return M + shift * np.eye(n)
        S, H = band(2.0 * b + 4.0), band(0.0)
        cnt = [0]
        A = ps.crawford_split(S, H, b, n // 2, cnt)
        N = n // b
        assert cnt[0] == 2 * ((N // 2 - 1) * (N // 2 - 2) // 2) + (N // 2 - 1) and cnt[0] < 0.62 * ((N - 1) * (N - 2) // 2)
        i, j = np.indices(A.shape)
        scale = np.max(np.abs(A))
        assert np.max(np.abs(A[np.abs(i - j) > 2 * b - 1])) == 0.0
        inner = (np.abs(i - j) > b) & (np.minimum(i, j) >= b) & (np.maximum(i, j) < n - b)
        assert np.max(np.abs(A[inner])) <= 1e-14 * scale
        ref = sla.eigh(H, S, eigvals_only=True)
        assert np.max(np.abs(np.linalg.eigvalsh((A + A.T) / 2) - ref)) <= 1e-13 * np.max(np.abs(ref))
