#!/usr/bin/env python3
"""bench.py -- l-channel eigensolves/sec at N_bsp = 4096 (BASELINE.json metric) on N MI355X.

One "step" = one pass of the hot path over one batch of l-channels per GPU, inputs resident in HBM:
  assembly of S and H(l) bands -> reduction to a band matrix -> tridiagonal -> bisection (all nfun eigenvalues of every
  channel) + the (l_ini, n0_ini) eigenvector and its WRITE_WF table on the rank that owns l_ini + the RCCL all-gather of the
  spectra.  The library's default route for k <= 9 is the BAND route (csrc/crawford.hip: the pencil stays banded at half-width 8,
  one-column chase on tiles of 8); the DENSE route (banded Cholesky -> standard form -> sy2sb -> two-step bulge chasing: north_star's letter) is measured
  in the same run as `dense_two_stage` (a few extra untimed-for-`value` steps) and is what `--route 1` makes `value`.
Workload (BASELINE configs[3], "Hydrogen l=0..127, N_bsp=4096"): KIND_GRID=0 ra=0 rb=800 k=9
nfun=4096 Zatom=1.  The l-loop being sharded is reference matrices.f90:242-248.

  python bench.py --gpus N --steps K --warmup W

N = 1 runs in this process.  N > 1 without RANK in the environment: this process starts
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py ...`
as a CHILD (subprocess, before anything here has touched a GPU or imported torch), relays the child's one JSON
line and exits with its code.  Under a launcher (RANK set) every rank runs `run()`.

`value`: at N = 1 the 128 channels of BASELINE configs[3] on the one GPU.  At N > 1 the default is `--scaling strong`:
BASELINE configs[3] exactly as written -- 128 channels in total, 128/N per GPU (round-3 verdict: the honest headline of an
N-GPU line); the line also carries `weak_scaling` (every GPU solves `--channels` consecutive channels, rank r takes
l = r*channels .. (r+1)*channels-1; cost per channel does not depend on l), measured in the same run (K more steps after the
timed region of `value`, same barrier / max-over-ranks bracket).  `--scaling weak` swaps the two.  The shards come from bspatom_amd/parallel.py (channel_range, gather_spectra -- the
code the gloo world-size-2 tests cover); no data-path collective except the final gather.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (whole path against the fp64 peak; per
kernel: launch durations measured in THIS run with HIP events around every launch, in one extra untimed step) and
`cpu_baseline` (the compiled reference oracle/_ref/ref_dump.x when present, else the CPU oracle port).
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_PEAK_TFLOPS = 78.6     # MI355X fp64 vector = matrix peak (datasheet)
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def cpu_baseline(sample_nfun, k):
    """Time the reference path (MATRIX_SVT + SOLVE_SYSTEM, one l-channel) on the host cores."""
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = min(avail, 16)            # one GPU's CPU share on the bench box; pins the BLAS thread count
    env = dict(os.environ, OPENBLAS_NUM_THREADS=str(cores), OMP_NUM_THREADS=str(cores))
    refx = os.path.join(ROOT, "oracle", "_ref", "ref_dump.x")
    rb = 800.0 * sample_nfun / 4096.0
    nml = ("&VARS_BSP KIND_GRID=0 ra=0.0D0 rb=%.1fD0 k=%d nfun=%d &end\n"
           "&VARS_TISE n0_ini=1 l_ini=0 l_fin=0 Zatom=1.0D0 &end\n&VARS_FIELD KIND_PI=0 &end\n" % (rb, k, sample_nfun))
    if os.path.exists(refx):
        try:
            with tempfile.TemporaryDirectory(prefix="bspbench.") as tmp:
                t0 = time.time()
                p = subprocess.run([refx], input=nml, cwd=tmp, capture_output=True, text=True, timeout=900, env=env)
                wall = time.time() - t0
            tim = {l.split()[0]: float(l.split()[1]) for l in p.stdout.split("\n") if l.startswith("REF_TIME_")}
            if p.returncode == 0 and "REF_TIME_SOLVE_SYSTEM_S" in tim:
                t = tim["REF_TIME_MATRIX_SVT_S"] + tim["REF_TIME_SOLVE_SYSTEM_S"]
                res = {"value": 1.0 / t, "unit": "eigensolves/s", "cores": cores, "kind": "reference",
                       "sample": "1 l-channel at nfun=%d k=%d, measured (compiled reference: MATRIX_SVT %.2fs + SOLVE_SYSTEM/"
                                 "DSYGV('V') %.2fs, flang -O2, OpenBLAS LAPACK 3.12, %d threads; wall %.1fs)"
                                 % (sample_nfun, k, tim["REF_TIME_MATRIX_SVT_S"], tim["REF_TIME_SOLVE_SYSTEM_S"], cores, wall)}
                if sample_nfun != 4096:
                    res["value_scaled_to_nfun4096"] = (1.0 / t) * (sample_nfun / 4096.0) ** 3
                return res
        except Exception as e:     # fall through to the port
            sys.stderr.write("cpu_baseline: reference binary failed (%s), using the oracle port\n" % e)
    os.environ["OPENBLAS_NUM_THREADS"] = str(cores)
    import oracle as orc
    c = orc.make_cfg(kind_grid=0, ra=0.0, rb=rb, k=k, nfun=sample_nfun, l_fin=0, zatom=1.0)
    t0 = time.time()
    orc.solve_all(c)
    t = time.time() - t0
    return {"value": 1.0 / t, "unit": "eigensolves/s", "cores": cores, "kind": "port",
            "sample": "1 l-channel at nfun=%d k=%d (oracle: banded assembly + scipy LAPACK dsygv 'V'); nfun=4096 costs %.1fx more"
                      % (sample_nfun, k, (4096.0 / sample_nfun) ** 3),
            "value_scaled_to_nfun4096": (1.0 / t) * (sample_nfun / 4096.0) ** 3}


def kernel_sources_sha():
    """sha256 over the kernel sources (csrc/*.hip, common.h): committed counter profiles carry it (profiles/*.json,
    key "csrc_sha16") so that a profile taken with other kernels is recognised as stale."""
    import glob
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "bspatom_amd", "csrc", "*.hip")) + [os.path.join(ROOT, "bspatom_amd", "csrc", "common.h")]):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def profile_summary(channels, nfun, route=1):
    """The newest committed counter profiles for this workload: profiles/*_pmc_summary.json (tools/pmc_summary.py: two separate
    rocprofv3 --pmc passes, FETCH_SIZE x2 gfx950 correction, WRITE_SIZE exact) and profiles/*_mfma_util.json (MFMA pipe busy,
    kernel alone).  rocprofv3 cannot run inside this process: these are COMMITTED PROFILES, named in the line, and flagged
    `stale` when their csrc_sha16 is not the hash of the kernel sources this run was built from.  Kernel DURATIONS do not
    come from here: they are measured live (bspatom_kernel_times)."""
    import glob
    sha = kernel_sources_sha()

    def newest(pat, ok):
        for f in sorted(glob.glob(os.path.join(ROOT, "profiles", pat)), reverse=True):
            try:
                d = json.load(open(f))
            except Exception:
                continue
            if ok(d):
                return d, os.path.relpath(f, ROOT), d.get("csrc_sha16") != sha
        return None, None, None
    # a profile belongs to a route by the kernels it holds (the band route's has crawford_item_kernel, the dense route's the GEMMs)
    mine = lambda d: any("crawford_item" in k for k in d.get("kernels", {})) == (route == 2)
    pmc, pmc_file, pmc_stale = newest("*_pmc_summary.json", lambda d: d.get("workload", {}).get("channels") == channels and
                                      d.get("workload", {}).get("nfun") == nfun and mine(d))
    mf, mf_file, mf_stale = newest("*_mfma_util.json", lambda d: "kernels" in d and mine(d))
    for f, st in ((pmc_file, pmc_stale), (mf_file, mf_stale)):
        if f and st:
            sys.stderr.write("bench: committed profile %s was taken with other kernel sources (csrc_sha16 differs): its "
                             "figures are marked stale in the line; re-run tools/gpu_profiles.sh\n" % f)
    return pmc, pmc_file, pmc_stale, ({k.replace("bsp::", ""): v for k, v in mf["kernels"].items()} if mf else {}), mf_file, mf_stale


def syr2k_tiles(npad):
    """128 x 128 tiles the rank-128 update executes per channel, summed over the panels (gemm_f64.hip::syr2k_lower_f64:
    row block bx holds column blocks 0 .. min(bx + 1, nb - 1))."""
    tiles = 0
    for p in range(npad // 64 - 1):
        m = npad - (p + 1) * 64
        nb = (m + 127) // 128
        tiles += sum(min(bx + 1, nb - 1) + 1 for bx in range(nb))
    return tiles


def symm_tiles(npad):
    """(row tiles of 128, K) pairs of symm per channel: sum over panels of ceil(m/128) * m."""
    return sum(((npad - (p + 1) * 64 + 127) // 128) * (npad - (p + 1) * 64) for p in range(npad // 64 - 1))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(args, argv):
    """--gpus N > 1 and no launcher around us: start the N ranks as a child `torch.distributed.run` (never exec, and
    nothing in this process has touched the GPU: torch is not even imported), relay its JSON line, return its code."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + argv
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    sys.stderr.write("bench: starting %d ranks: %s\n" % (args.gpus, " ".join(cmd)))
    p = subprocess.Popen(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, text=True)
    out, _ = p.communicate()
    lines = [l for l in out.split("\n") if l.startswith("{")]
    for l in out.split("\n"):
        if l and not l.startswith("{"):
            sys.stderr.write(l + "\n")           # launcher chatter does not belong on the one-line stdout
    if lines:
        sys.stdout.write(lines[-1] + "\n")
        sys.stdout.flush()
    if p.returncode == 0 and len(lines) != 1:
        sys.stderr.write("bench: expected one JSON line from the ranks, got %d\n" % len(lines))
        return 1
    return p.returncode


def run(args):
    selftest = args.selftest_launcher
    import torch                      # first: its HIP runtime is the one the process uses
    import torch.distributed as dist
    import numpy as np
    from bspatom_amd import parallel

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench: --gpus %d but WORLD_SIZE %d (start the ranks with --nproc-per-node %d, or run "
                         "`python bench.py --gpus %d` without a launcher and let it start them)" % (args.gpus, world, args.gpus, args.gpus))
    use_dist = "RANK" in os.environ and "MASTER_ADDR" in os.environ     # launched by torch.distributed.run (also at N=1)
    if selftest:
        # CPU rehearsal of the launch path (tests/test_host_cpu.py): gloo, no GPU, no solve -- the spectra are stand-ins that
        # encode (channel, index); everything else (sharding, gather, barrier bracket, max over ranks, one line) is the real code
        dev = torch.device("cpu")
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X: libbspatom has no CPU path")
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
    # rank 0 prints ONE line on stdout: RCCL writes its banner (version, hostname, library path) to stdout when the
    # communicator is created, so everything but the result line goes to stderr
    real_stdout = os.dup(1)
    sys.stdout.flush()
    os.dup2(2, 1)
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if selftest:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=dev)

    def shard(mode):
        if mode == "weak":
            total = world * args.channels
            cnts = [parallel.channel_range(r, world, total - 1, per_rank=args.channels)[1] for r in range(world)]
            l0_, nl_ = parallel.channel_range(rank, world, total - 1, per_rank=args.channels)
        else:
            total = args.channels
            cnts = [parallel.channel_range(r, world, total - 1)[1] for r in range(world)]
            l0_, nl_ = parallel.channel_range(rank, world, total - 1)
        return total, cnts, l0_, nl_

    modes = [args.scaling] + ([m for m in ("weak", "strong") if m != args.scaling] if world > 1 else [])
    l_ini = 0
    total_max = max(shard(m)[0] for m in modes)
    if selftest:
        prob, n = None, args.nfun
    else:
        from bspatom_amd import capi
        inp = capi.make_input(kind_grid=0, ra=0.0, rb=args.rb, k=args.k, nfun=args.nfun, n0_ini=1, l_ini=l_ini,
                              l_fin=total_max - 1, zatom=1.0)
        prob = capi.Problem(inp, device=local)
        n = prob.nfun
        if args.route:
            capi.set_option("route", args.route)
    E_dev = torch.empty(max(max(shard(m)[3] for m in modes), 1) * n, dtype=torch.float64, device=dev)

    def sync():
        if use_dist:
            dist.barrier()
        if not selftest:
            torch.cuda.synchronize()

    def measure(mode, steps, warmup):
        """W untimed + K timed steps of one sharding; returns the per-mode record (rank 0 keeps E_all for the checks)."""
        total, counts, l0, nl = shard(mode)
        owner = next(r for r in range(world) if sum(counts[:r]) <= l_ini < sum(counts[:r + 1]))
        stage_ms = np.zeros(6)
        state = {"E_all": None}

        def step(timed):
            if nl > 0 and not selftest:
                info = prob.solve_dev(l0, nl, E_dev.data_ptr())      # returns when the library's stream has drained
                assert (info == 0).all()
                if timed:
                    t = prob.last_timing()
                    stage_ms[:] += [t["assemble"], t["chol_std"], t["sy2sb"], t["sb2st"], t["bisect"], t["total"]]
            elif nl > 0:
                E_dev[: nl * n] = (torch.arange(l0, l0 + nl, dtype=torch.float64).repeat_interleave(n) * 1e6 +
                                   torch.arange(n, dtype=torch.float64).repeat(nl))
            if rank == owner and not selftest:     # owner of l_ini: the one eigenvector KIND_PI=0 consumes + WRITE_WF
                c = prob.eigvec(l_ini, 1)
                prob.write_wf(c)
            # all-gather of the spectra: the only collective of the path (RCCL; at N = 1 under a launcher too).  E_dev is
            # rewritten by the next solve on the library's own stream, so the gather must have completed before the step ends.
            state["E_all"] = parallel.gather_spectra(E_dev[: nl * n], n, counts)
            if use_dist and not selftest:
                torch.cuda.current_stream().synchronize()

        for _ in range(warmup):
            step(False)
        sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            step(True)
        sync()
        dt = time.perf_counter() - t0
        if use_dist:
            tt = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return {"mode": mode, "total": total, "counts": counts, "l0": l0, "nl": nl, "owner": owner, "dt": dt,
                "stage_ms": stage_ms / max(steps, 1), "E_all": state["E_all"],
                "value": total * steps / dt, "ms_per_step": 1e3 * dt / steps}

    recs = [measure(m, args.steps, args.warmup) for m in modes]
    main = recs[0]
    total, counts, nl = main["total"], main["counts"], main["nl"]

    # one more, UNTIMED step of the main sharding with HIP events around every launch of the big kernels (option "ktime")
    ktimes = None
    kstage = None
    if not selftest and nl > 0 and not args.no_kernel_timing:
        capi.set_option("ktime", 1)
        prob.solve_dev(main["l0"], nl, E_dev.data_ptr())
        ktimes = capi.kernel_times()
        capi.set_option("ktime", 0)
        kstage = prob.last_timing()
    sync()

    # two extra legs, untimed for `value` (N = 1 only): the dense route (north_star's letter) when `value` came from the band route,
    # and one call of bsp_dsygv_('V') -- all eigenvectors, the contract of the reference's call site (matrices.f90:248)
    route = prob.route() if not selftest else 0
    dense_leg = fullv_leg = None
    if not selftest and world == 1 and nl > 0:
        if route == 2 and not args.no_dense_leg:
            dense_leg = dense_route_leg(args, prob, capi, torch, main["l0"], nl, E_dev, n)
        if not args.no_full_v:
            fullv_leg = full_v_leg(args, prob, capi, n)
    sync()

    if rank == 0:
        out = {"metric": "l-channel eigensolves/sec at N_bsp=%d fp64" % n, "value": main["value"], "unit": "eigensolves/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": main["ms_per_step"],
               "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic"}
        for rec in recs:
            E_all = rec["E_all"]
            assert tuple(E_all.shape) == (rec["total"], n), (tuple(E_all.shape), rec["total"], n)
            if selftest:
                want = (torch.arange(rec["total"], dtype=torch.float64)[:, None] * 1e6 + torch.arange(n, dtype=torch.float64)[None, :])
                assert torch.equal(E_all.cpu(), want), "gathered stand-in spectra are not in channel order"

        def describe(rec):
            per = "/".join(str(c) for c in sorted(set(rec["counts"]), reverse=True))
            return ("%d l-channels per GPU, %d in total" % (args.channels, rec["total"])) if rec["mode"] == "weak" else \
                   ("%d l-channels in total, %s per GPU" % (rec["total"], per))
        out["config"] = {
            "workload": "Hydrogen Coulomb l=%d..%d, N_bsp=%d, k=%d, KIND_GRID=0 rb=%g (BASELINE configs[3]); `value` = %s scaling: %s"
                        % (0, total - 1, n, args.k, args.rb, args.scaling, describe(main)),
            "channels_per_gpu": counts, "channels_total": total, "eigenvector_owner_rank": main["owner"],
            "parallelism": "l-sharded x%d (bspatom_amd/parallel.py), %s all-gather of spectra"
                           % (world, ("gloo (launcher self-test, no solve)" if selftest else "RCCL") if use_dist else "no (single process)"),
            "collective_calls": parallel.COLLECTIVE_CALLS, "launched_by": "torch.distributed.run" if use_dist else "python"}
        if selftest:
            out["data"] = "launcher self-test: stand-in spectra, no solve"
        for rec in recs[1:]:
            key = "configs3_as_stated" if rec["mode"] == "strong" else "weak_scaling"
            out[key] = {"scaling": rec["mode"], "value": rec["value"], "unit": "eigensolves/s", "ms_per_step": rec["ms_per_step"],
                        "steps": args.steps, "warmup": args.warmup, "workload": describe(rec),
                        "channels_per_gpu": rec["counts"], "channels_total": rec["total"],
                        "stage_ms_per_step_rank0": dict(zip(["assemble", "chol_std", "sy2sb", "sb2st", "bisect", "total_device"],
                                                            [float(x) for x in rec["stage_ms"]]))}
        if not selftest:
            out.update(report(args, world, n, prob.npad, main, ktimes, kstage, route))
            if dense_leg:
                out["dense_two_stage"] = dense_leg
            if fullv_leg:
                out["full_V"] = fullv_leg
            if not args.no_cpu_baseline and world == 1:       # reported at N=1 only (rank 0's host cores)
                out["cpu_baseline"] = cpu_baseline(args.cpu_sample_nfun, args.k)
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if prob is not None:
        prob.close()


STAGES_DENSE = ["assemble", "chol_std", "sy2sb", "sb2st", "bisect"]
STAGES_BAND = ["assemble", "(unused)", "band_reduction", "band_chase", "bisect"]


def flop_dense(n, k):
    """algorithmic flop per l-channel of the dense two-stage route"""
    return 4.0 / 3.0 * n ** 3 + 4.0 * n ** 2 * k


def flop_band(n):
    """Algorithmic flop per l-channel of the band route (csrc/crawford.hip + one-column chase + bisection), stated in DESIGN.md 4.5:
    chase items (N-1)(N-2)/2 with N = ceil(n/8), each the RQ of an 8 x 16 block (2*16*64 - 2/3*512), its 16 x 16 factor formed
    (8 reflectors x 4*256), the congruence Q^T (W Q) (2 products of 2*16^3) and Q^T [E; 0] (2*16*8*8); N - 1 eliminations (the two
    products and the two small ones); band of half-width 8 -> tridiagonal: 6 n^2 8; bisection: 54 Sturm counts of n rows for n
    eigenvalues, 3 fp64 operations per row -- until round 4's secant rounds (32 evaluation rounds of a workgroup's 1024 slots) and
    shared points: now ~22 rounds (~15 lock-step rounds until an eighth of the brackets are left, ~7 of the multisection tail;
    the kernel's VALU instruction count / that of one round, profiles/r04_band_valu_util.txt)."""
    N = (n + 7) // 8
    item = (2 * 16 * 64 - 2.0 / 3.0 * 512) + 8 * 4 * 256 + 2 * 2 * 16 ** 3 + 2 * 16 * 8 * 8
    elim = 2 * 2 * 16 ** 3 + 2 * 2 * 16 * 8 * 8
    return {"band_reduction": (N - 1) * (N - 2) / 2.0 * item + (N - 1) * elim, "band_chase": 6.0 * n * n * 8,
            "bisect": 22.0 * 3.0 * n * n}


def dense_kernel_entries(ktimes, npad, nl, pmc_bytes, pmc_file, pmc_stale, mfma, mfma_file, mfma_stale):
    """the two big products of sy2sb against the MFMA peak.  `frac` = executed flop / the SUM of the kernel's launch durations.
    Launches of the two channel groups (and of the look-ahead) overlap on the chip, so every launch shares the CUs with others
    while it runs: the figure is what the kernel achieves IN the pipeline (a lower bound of what it reaches alone;
    `mfma_pipe_busy` is the counter figure of the kernel running alone, from the committed profile)."""
    kern = []
    src_live = "HIP events around every launch, one extra untimed step of this run (bspatom_kernel_times)"

    def kt(sub):
        hit = [(k, v) for k, v in (ktimes or {}).items() if sub in k]
        return hit[0][1] if hit else (0.0, 0)
    for sub, flop, label, prof in (("syr2k", syr2k_tiles(npad) * 2.0 * 128 ** 3 * nl, "rank-128 update A22 -= [V Z][Z V]^T (syr2k)", "gemm2_kernel<128, 128"),
                                   ("symm", symm_tiles(npad) * 2.0 * 64 * 128 * nl, "symm Y = A22 W", "gemm2_kernel<64, 128")):
        ms_sum, calls = kt(sub)
        if calls == 0:
            continue
        ach = flop / (ms_sum * 1e-3) / 1e12
        util = next((v["mfma_util"] for k, v in mfma.items() if k.startswith(prof)), None)
        kern.append({"kernel": next(k for k in ktimes if sub in k), "what": label, "bound": "mfma", "launches_per_step": calls,
                     "avg_launch_ms": ms_sum / calls, "kernel_ms_per_step": ms_sum, "launch_ms_source": src_live,
                     "flop_per_step": flop, "achieved": ach, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / FP64_PEAK_TFLOPS,
                     "frac_definition": "executed flop / sum of this kernel's launch durations (launches overlap with other kernels of the pipeline)",
                     "mfma_pipe_busy": util, "mfma_pipe_busy_source": mfma_file, "mfma_pipe_busy_stale": mfma_stale,
                     "traffic": pmc_bytes(prof), "traffic_source": pmc_file, "traffic_stale": pmc_stale})
    other = {"panel_qr": dict(zip(("kernel_ms_per_step", "launches_per_step"), kt("panel_qr"))),
             "sy2sb_chain_small_products": dict(zip(("kernel_ms_per_step", "launches_per_step"), kt("sy2sb chain"))),
             "sb2sb_mfma_kernel": dict(zip(("kernel_ms_per_step", "launches_per_step"), kt("sb2sb_mfma"))),
             "sbr_rows_kernel<16>": dict(zip(("kernel_ms_per_step", "launches_per_step"), kt("sbr_rows"))),
             "bisect3_kernel": dict(zip(("kernel_ms_per_step", "launches_per_step"), kt("bisect3"))),
             "cholesky_std_form": dict(zip(("kernel_ms_per_step", "launches_per_step"), kt("std_form")))}
    return kern, other


def dense_route_leg(args, prob, capi, torch, l0, nl, E_dev, n):
    """north_star's letter -- banded Cholesky, standard form, two-stage tridiagonalisation with MFMA panel-update GEMMs -- measured in
    the same run on the same channels: 1 warm-up + 2 timed solves (not part of `value`), then one step with per-launch events."""
    capi.set_option("route", 1)
    try:
        prob.solve_dev(l0, nl, E_dev.data_ptr())
        torch.cuda.synchronize()
        st = [0.0] * 6
        t0 = time.perf_counter()
        for _ in range(2):
            prob.solve_dev(l0, nl, E_dev.data_ptr())
            t = prob.last_timing()
            st = [a + b / 2 for a, b in zip(st, [t["assemble"], t["chol_std"], t["sy2sb"], t["sb2st"], t["bisect"], t["total"]])]
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 2
        kt = None
        if not args.no_kernel_timing:
            capi.set_option("ktime", 1)
            prob.solve_dev(l0, nl, E_dev.data_ptr())
            kt = capi.kernel_times()
            capi.set_option("ktime", 0)
    finally:
        capi.set_option("route", args.route)
    pmc, pmc_file, pmc_stale, mfma, mfma_file, mfma_stale = profile_summary(nl, n, route=1)

    def pmc_bytes(kname):
        if not pmc:
            return None
        return next((v["traffic_bytes_per_launch"] for k, v in pmc.get("kernels", {}).items() if kname in k), None)
    kern, other = dense_kernel_entries(kt, prob.npad, nl, pmc_bytes, pmc_file, pmc_stale, mfma, mfma_file, mfma_stale)
    val = nl / dt
    ach = flop_dense(n, args.k) * val / 1e12
    return {"what": "the dense route (BSP_ROUTE=1): banded Cholesky -> standard form (dense C_l) -> sy2sb with MFMA_F64 panel-update GEMMs -> "
                    "band 64 -> 16 -> 1 -> bisection; same channels, 1 warm-up + 2 timed solves in this run, no eigenvector / gather",
            "value": val, "unit": "eigensolves/s", "ms_per_step": 1e3 * dt,
            "stage_ms_per_step": dict(zip(STAGES_DENSE + ["total_device"], st)),
            "roofline": {"bound": "mfma", "achieved": ach, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / FP64_PEAK_TFLOPS,
                         "scope": "F(n) = 4/3 n^3 + 4 n^2 k flop per l-channel (SURVEY 8d) x eigensolves/s against the fp64 peak",
                         "kernels": kern, "other_kernels": other}}


def full_v_leg(args, prob, capi, n):
    """One call of bsp_dsygv_('V') on channel l = 0 of the workload: all n eigenvalues AND all n S-orthonormal eigenvectors through
    the Fortran-77 symbol boundary, dense A and B in, as the reference's call site has it (matrices.f90:248) -- the unit of
    work the CPU baseline times.  SURVEY 8(d): a run that computes all eigenvectors reports F_V separately."""
    import numpy as np
    SB, HB = prob.assemble(0, 1)
    k = SB.shape[0]
    A = np.zeros((n, n), order="F"); B = np.zeros((n, n), order="F")
    for d in range(k):
        i = np.arange(n - d)
        A[i, i + d] = HB[0, d, :n - d]; B[i, i + d] = SB[d, :n - d]
    capi.dsygv(A, B, "N")                      # warm-up of the device pools
    t0 = time.perf_counter()
    w, Z, _, info = capi.dsygv(A, B, "V")
    dt = time.perf_counter() - t0
    assert info == 0
    cols = np.linspace(0, n - 1, 64).astype(int)           # a sample of the contract: residual and S-orthonormality of 64 vectors
    Hd = A + A.T - np.diag(np.diag(A)); Sd = B + B.T - np.diag(np.diag(B))
    Zs = Z[:, cols]
    res = np.max(np.abs(Hd @ Zs - (Sd @ Zs) * w[cols])) / np.max(np.abs(w))
    orth = np.max(np.abs(Zs.T @ (Sd @ Zs) - np.eye(len(cols))))
    FV = flop_dense(n, args.k) + 4.0 * n ** 3
    return {"what": "bsp_dsygv_(1, 'V', 'U') on channel l = 0 (n = %d): all eigenvalues and all S-orthonormal eigenvectors, host arrays in and "
                    "out (the symbol boundary, SURVEY 8b.2); one call after a warm-up call" % n,
            "value": 1.0 / dt, "unit": "channels/s", "seconds_per_channel": dt,
            "F_V_flop_per_channel": FV, "F_V_definition": "F(n) + 4 n^3 (SURVEY 8d: the dense algorithm's count for JOBZ = 'V')",
            "dense_equivalent_tflops": FV / dt / 1e12,
            "check_64_vectors": {"max_residual_over_lambda_max": float(res), "max_S_orthonormality_defect": float(orth)}}


def report(args, world, n, npad, main, ktimes, kstage, route):
    """roofline (whole path + per kernel), stage times and the Rydberg check of rank 0's spectra."""
    import numpy as np
    from bspatom_amd import capi
    total, nl = main["total"], main["nl"]
    E_all = main["E_all"]
    Eh = E_all[0].cpu().numpy()
    ryd = max(abs(Eh[i] + 0.5 / (i + 1) ** 2) / (0.5 / (i + 1) ** 2) for i in range(8))
    Elast = E_all[total - 1].cpu().numpy()                      # a channel the last rank solved
    assert np.all(np.diff(Elast) >= 0) and (total == 1 or Elast[0] > Eh[0])
    stage_ms = main["stage_ms"]
    pmc, pmc_file, pmc_stale, mfma, mfma_file, mfma_stale = profile_summary(nl, n, route=route)

    def pmc_bytes(kname):
        if not pmc:
            return None
        return next((v["traffic_bytes_per_launch"] for k, v in pmc.get("kernels", {}).items() if kname in k), None)

    def kt(sub):
        hit = [(k, v) for k, v in (ktimes or {}).items() if sub in k]
        return hit[0][1] if hit else (0.0, 0)
    src_live = "HIP events around every launch, one extra untimed step of this run (bspatom_kernel_times)"
    pmc_total = (sum(v["traffic_bytes_per_launch"] * v["launches"] for v in pmc["kernels"].values()) if pmc else None)
    traffic_note = ("HBM bytes of ONE STEP (all kernels: sum of launches x bytes per launch), from the committed profile %s%s -- not measured "
                    "in this run" % (pmc_file, " (STALE: taken with other kernel sources)" if pmc_stale else "")) if pmc else \
                   "no committed counter profile of this route and workload"
    if route == 2:
        # ---- band route: three kernels of comparable weight.  The reduction and the bisection compute on the fp64 vector / matrix
        # datapath (one peak on gfx950: 78.6 TFLOP/s), the one-column chase streams the band through LDS windows (HBM model).
        fb = flop_band(n)
        N = (n + 7) // 8
        kern = []
        ms_sum, calls = kt("crawford")
        if calls:
            ach = fb["band_reduction"] * nl / (ms_sum * 1e-3) / 1e12
            items = (N - 1) * (N - 2) // 2
            grp = max(1, min(int(capi.get_option("cw_streams")), 4))          # channel groups, each on a stream of its own
            while grp > 1 and nl // grp < 16:
                grp -= 1
            kern.append({"kernel": "crawford_item4_kernel (+ its set-up kernels; crawford_item_kernel with BSP_CW_ITEMS4=0; csrc/crawford.hip)", "what": "banded pencil -> band of half-width 8: "
                         "%d chase items of 8 x 8 blocks per channel in %d wavefronts = %d launches (the channels in %d groups, each on a stream of its own)"
                         % (items, 3 * N - 5, (3 * N - 5) * grp, grp), "bound": "mfma",
                         "launches_per_step": (3 * N - 5) * grp + 6, "kernel_ms_per_step": ms_sum, "avg_launch_ms": ms_sum / ((3 * N - 5) * grp),
                         "avg_launch_ms_definition": "the stage's time / its launches; the groups' launches overlap, a launch alone takes ~2 x this",
                         "launch_ms_source": "HIP events around the whole stage, one extra untimed step of this run (bspatom_kernel_times)",
                         "flop_per_step": fb["band_reduction"] * nl, "achieved": ach, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": ach / FP64_PEAK_TFLOPS, "bytes_model_per_step": items * 11 * 512.0 * nl,
                         "bytes_model_definition": "11 blocks of 512 B read or written per item (mostly L2 / Infinity Cache hits: the working "
                                                   "set of a channel is 0.8 MB)",
                         "traffic": pmc_bytes("crawford_item"), "traffic_source": pmc_file, "traffic_stale": pmc_stale,
                         "limited_by": "the life of ONE wave of four items (23 500 cycles: eight dependent reflector steps, 48 fp64 MFMAs at 64 cycles) "
                                       "times the rounds of waves a wavefront launch needs: a launch has 2750 waves on average and the chip "
                                       "holds 2048 (two per SIMD, 250 registers); DESIGN.md 4.5, profiles/r04_experiments.txt 14 - 16"})
        ms_sum, calls = kt("sbr_rows")
        # tiles of 8: a pass of 16 sweeps streams the remaining band (window columns of 16 rows) in and out once
        b2 = sum(2 * 16 * 8 * (n - s0_) for s0_ in range(0, n - 2, 16)) * nl
        if calls:
            kern.append({"kernel": next(k for k in ktimes if "sbr_rows" in k), "what": "band of half-width 8 -> tridiagonal, tiles of 8 "
                         "(sbr_rows_kernel<8>)", "bound": "hbm", "launches_per_step": calls, "avg_launch_ms": ms_sum / calls,
                         "kernel_ms_per_step": ms_sum, "launch_ms_source": src_live, "bytes_model_per_step": b2,
                         "achieved": b2 / (ms_sum * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": b2 / (ms_sum * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "traffic": pmc_bytes("sbr_rows"), "traffic_source": pmc_file, "traffic_stale": pmc_stale,
                         "limited_by": "serial chase, one item per sweep and step: the LDS pipe (~176 LDS instructions per step of a workgroup, two workgroups per CU) "
                                       "and the latency of an item's dependent chains"})
        ms_sum, calls = kt("bisect3")
        if calls:
            ach = fb["bisect"] * nl / (ms_sum * 1e-3) / 1e12
            kern.append({"kernel": "bisect3_kernel", "bound": "mfma", "launches_per_step": calls, "avg_launch_ms": ms_sum / calls,
                         "kernel_ms_per_step": ms_sum, "launch_ms_source": src_live, "flop_per_step": fb["bisect"] * nl, "achieved": ach,
                         "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / FP64_PEAK_TFLOPS,
                         "limited_by": "fp64 vector pipe: 3 fp64 + ~1.6 other instructions per row and eigenvalue; ~12 - 15 evaluations per eigenvalue, "
                                       "~22 rounds per workgroup in lock step (DESIGN 4.3)"})
        F = sum(fb.values())
        ach = F * main["value"] / 1e12
        ceiling = FP64_PEAK_TFLOPS * 1e12 * world / flop_dense(n, args.k)
        roof = {"bound": "mfma", "achieved": ach, "peak": FP64_PEAK_TFLOPS * world, "unit": "TFLOP/s", "frac": ach / (FP64_PEAK_TFLOPS * world),
                "scope": "whole path, BAND route: F_band(n) = %.3g flop per l-channel (band reduction %.3g + one-column chase %.3g + bisection %.3g; "
                         "bench.py::flop_band, DESIGN.md 4.5) x eigensolves/s, against the fp64 vector = matrix peak of %d GPU(s).  The route "
                         "does 1/%.0f of the dense algorithm's arithmetic (SURVEY 8d: F(n) = %.3g): see `dense_algorithm_ceiling`"
                         % (F, fb["band_reduction"], fb["band_chase"], fb["bisect"], world, flop_dense(n, args.k) / F, flop_dense(n, args.k)),
                "dense_algorithm_ceiling": {"eigensolves_per_s_at_100_percent_of_fp64_peak": ceiling,
                                            "value_over_ceiling": main["value"] / ceiling,
                                            "note": "F(n) = 4/3 n^3 + 4 n^2 k per channel: no implementation of the dense two-stage route can "
                                                    "exceed this rate on %d GPU(s); `dense_two_stage` is that route measured in this run" % world},
                "traffic": pmc_total, "traffic_note": traffic_note, "kernels": kern, "kernel_timing_step_stage_ms": kstage}
        names = STAGES_BAND
    else:
        kern, other = dense_kernel_entries(ktimes, npad, nl, pmc_bytes, pmc_file, pmc_stale, mfma, mfma_file, mfma_stale)
        ach = flop_dense(n, args.k) * main["value"] / 1e12
        roof = {"bound": "mfma", "achieved": ach, "peak": FP64_PEAK_TFLOPS * world, "unit": "TFLOP/s",
                "frac": ach / (FP64_PEAK_TFLOPS * world),
                "scope": "whole path, DENSE route: F(n) = 4/3 n^3 + 4 n^2 k flop per l-channel (SURVEY 8d) x eigensolves/s, against the fp64 "
                         "matrix/vector peak of %d GPU(s)" % world,
                "traffic": pmc_total, "traffic_note": traffic_note, "kernels": kern, "other_kernels_this_run": other,
                "kernel_timing_step_stage_ms": kstage}
        names = STAGES_DENSE
    return {"roofline": roof, "route": {1: "dense", 2: "band"}[route],
            "stage_ms_per_step_rank0": dict(zip(names + ["total_device"], [float(x) for x in stage_ms])),
            "rydberg_max_rel_err_n<=8": ryd, "csrc_sha16": kernel_sources_sha()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--nfun", type=int, default=4096)
    ap.add_argument("--k", type=int, default=9)
    ap.add_argument("--rb", type=float, default=800.0)
    ap.add_argument("--channels", type=int, default=128,
                    help="l-channels per GPU (weak) and in total (strong: BASELINE configs[3] as stated)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default=None,
                    help="which sharding is `value`: strong (default at N > 1) = --channels in total, --channels/N per GPU = BASELINE "
                         "configs[3] as stated; weak (default at N = 1, where the two coincide) = every GPU solves --channels channels.  "
                         "At N > 1 the other one is measured too and reported beside it")
    ap.add_argument("--route", type=int, default=0, choices=[0, 1, 2],
                    help="0 = the library's default route (band route for k <= 9), 1 = dense route (north_star's letter), 2 = band route")
    ap.add_argument("--no-dense-leg", action="store_true", help="skip the extra measurement of the dense route (`dense_two_stage`)")
    ap.add_argument("--no-full-v", action="store_true", help="skip the extra measurement of bsp_dsygv_('V') (`full_V`)")
    ap.add_argument("--cpu-sample-nfun", type=int, default=4096)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true",
                    help="skip the extra untimed step that brackets every launch with HIP events (profiler runs: the trace should hold "
                         "the timed steps only)")
    ap.add_argument("--selftest-launcher", action="store_true",
                    help="CPU rehearsal of the N-rank launch path (gloo, stand-in spectra, no GPU, no solve): tests only")
    args = ap.parse_args()
    if args.scaling is None:
        args.scaling = "strong" if args.gpus > 1 else "weak"
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(spawn_ranks(args, sys.argv[1:]))
    run(args)


if __name__ == "__main__":
    main()
