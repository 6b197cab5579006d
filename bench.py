#!/usr/bin/env python3
"""bench.py -- l-channel eigensolves/sec at N_bsp = 4096 (BASELINE.json metric) on N MI355X.

One "step" = one pass of the hot path over one batch of l-channels per GPU, inputs resident in HBM:
  assembly of S and H(l) bands -> banded Cholesky -> standard form -> sy2sb -> sb2st -> bisection
  (all nfun eigenvalues of every channel) + the (l_ini, n0_ini) eigenvector and its WRITE_WF table
  on the rank that owns l_ini + the RCCL all-gather of the spectra.
Workload (BASELINE configs[3], "Hydrogen l=0..127, N_bsp=4096"): KIND_GRID=0 ra=0 rb=800 k=9
nfun=4096 Zatom=1.  --scaling weak (default): every GPU solves `--channels` (128) consecutive l-channels, rank r
takes l = r*channels .. (r+1)*channels-1 (cost per channel does not depend on l); --scaling strong: configs[3] as
stated, 128 channels in total, 128/N per GPU.  The shards come from bspatom_amd/parallel.py (channel_range,
gather_spectra -- the code the gloo world-size-2 tests cover); no data-path collective except the final gather.

  python bench.py --gpus 1 --steps 3 --warmup 1
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant stage,
HIP-event time measured live on the library's stream) and `cpu_baseline` (the compiled reference
oracle/_ref/ref_dump.x when present, else the CPU oracle port) objects.
"""
import argparse
import json
import os
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


def profile_summary(channels, nfun):
    """The newest committed PMC summary for this workload (profiles/*_pmc_summary.json, written by
    tools/pmc_summary.py from two separate rocprofv3 --pmc passes: FETCH_SIZE x2 gfx950 correction, WRITE_SIZE exact)
    and kernel-stats file (profiles/*_kernel_stats.csv, rocprofv3 --kernel-trace --stats of this same command).
    rocprofv3 cannot run inside this process: these are COMMITTED PROFILES, named in the line, not measurements of
    this run.  Returns (pmc dict or None, pmc file, {kernel: avg ms}, stats file)."""
    import csv
    import glob
    pmc, pmc_file = None, None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_summary.json")), reverse=True):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if d.get("workload", {}).get("channels") == channels and d.get("workload", {}).get("nfun") == nfun:
            pmc, pmc_file = d, os.path.relpath(f, ROOT)
            break
    stats, stats_file = {}, None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*final_kernel_stats.csv")), reverse=True):
        try:
            for r in csv.DictReader(open(f)):
                stats[r["Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("bsp::", "")] = (float(r["AverageNs"]) * 1e-6, int(r["Calls"]))
            stats_file = os.path.relpath(f, ROOT)
            break
        except Exception:
            continue
    mfma, mfma_file = {}, None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_mfma_util.json")), reverse=True):
        try:
            mfma = {k.replace("bsp::", ""): v for k, v in json.load(open(f))["kernels"].items()}
            mfma_file = os.path.relpath(f, ROOT)
            break
        except Exception:
            continue
    return pmc, pmc_file, stats, stats_file, mfma, mfma_file


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--nfun", type=int, default=4096)
    ap.add_argument("--k", type=int, default=9)
    ap.add_argument("--rb", type=float, default=800.0)
    ap.add_argument("--channels", type=int, default=128,
                    help="l-channels per GPU (--scaling weak) or in total (--scaling strong)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak: every GPU solves --channels channels (BASELINE configs[3] at N=1); strong: configs[3] as "
                         "stated, --channels = 128 in total, 128/N per GPU")
    ap.add_argument("--cpu-sample-nfun", type=int, default=4096)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch                      # first: its HIP runtime is the one the process uses
    import torch.distributed as dist
    import numpy as np
    from bspatom_amd import capi, parallel

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            sys.stderr.write("bench: --gpus %d but WORLD_SIZE %d; using WORLD_SIZE\n" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: libbspatom has no CPU path")
    torch.cuda.set_device(local)
    use_dist = "RANK" in os.environ and "MASTER_ADDR" in os.environ     # launched by torch.distributed.run (also at N=1)
    # rank 0 prints ONE line on stdout: RCCL writes its banner (version, hostname, library path) to stdout when the
    # communicator is created, so everything but the result line goes to stderr
    real_stdout = os.dup(1)
    sys.stdout.flush()
    os.dup2(2, 1)
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))

    # l-sharding (bspatom_amd/parallel.py -- the code the gloo world-size-2 tests cover)
    if args.scaling == "weak":
        total = world * args.channels
        counts = [parallel.channel_range(r, world, total - 1, per_rank=args.channels)[1] for r in range(world)]
        l0, nl = parallel.channel_range(rank, world, total - 1, per_rank=args.channels)
    else:
        total = args.channels
        counts = [parallel.channel_range(r, world, total - 1)[1] for r in range(world)]
        l0, nl = parallel.channel_range(rank, world, total - 1)
    l_ini = 0
    owner = next(r for r in range(world) if sum(counts[:r]) <= l_ini < sum(counts[:r + 1]))
    inp = capi.make_input(kind_grid=0, ra=0.0, rb=args.rb, k=args.k, nfun=args.nfun, n0_ini=1, l_ini=l_ini,
                          l_fin=total - 1, zatom=1.0)
    prob = capi.Problem(inp, device=local)
    n = prob.nfun
    E_dev = torch.empty(max(nl, 1) * n, dtype=torch.float64, device="cuda")
    E_all = None

    stage_ms = np.zeros(6)

    def step(timed):
        nonlocal E_all
        if nl > 0:
            info = prob.solve_dev(l0, nl, E_dev.data_ptr())      # returns when the library's stream has drained
            assert (info == 0).all()
            if timed:
                t = prob.last_timing()
                stage_ms[:] += [t["assemble"], t["chol_std"], t["sy2sb"], t["sb2st"], t["bisect"], t["total"]]
        if rank == owner:             # owner of l_ini: the one eigenvector KIND_PI=0 consumes + WRITE_WF
            c = prob.eigvec(l_ini, 1)
            prob.write_wf(c)
        # RCCL all-gather of the spectra: the only collective of the path.  E_dev is rewritten by the next solve on the
        # library's own stream, so the gather must have completed before the step ends.
        E_all = parallel.gather_spectra(E_dev[: nl * n], n, counts)
        if use_dist:
            torch.cuda.current_stream().synchronize()

    def sync():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(False)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    sync()
    dt = time.perf_counter() - t0
    if use_dist:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    if rank == 0:
        assert tuple(E_all.shape) == (total, n)
        Eh = E_all[0].cpu().numpy()
        ryd = max(abs(Eh[i] + 0.5 / (i + 1) ** 2) / (0.5 / (i + 1) ** 2) for i in range(8))
        Elast = E_all[total - 1].cpu().numpy()                      # a channel the last rank solved
        assert np.all(np.diff(Elast) >= 0) and Elast[0] > Eh[0]
        stage_ms /= max(args.steps, 1)
        units = total * args.steps
        value = units / dt
        b = 64
        F = 4.0 / 3.0 * n ** 3 + 4.0 * n ** 2 * args.k            # SURVEY 8(d) flops per l-channel
        names = ["assemble", "chol_std", "sy2sb", "sb2st", "bisect"]
        pmc, pmc_file, kstats, stats_file, mfma, mfma_file = profile_summary(nl, n)

        def pmc_bytes(kname):
            if not pmc:
                return None
            for k, v in pmc.get("kernels", {}).items():
                if kname in k:
                    return v["traffic_bytes_per_launch"]
            return None

        # per-kernel rooflines.  sb2st is ONE launch per step: its duration is this run's HIP-event time on the
        # library's stream.  The two big GEMM kernels of sy2sb are ~190 launches each, overlapped on several streams:
        # their durations are the kernel-trace averages of the committed profile of this same command.
        kern = []
        sb_ms = float(stage_ms[3])
        # bulge chasing, n^2/(2b) chase items per channel, each with one b x b block and one b x b symmetric block.
        # bytes_min: the band read once + d, e written (what SURVEY 8(d) calls algorithmic: the data the stage must
        # touch); bytes_pass_model: what the two-sweeps-per-pass scheme moves by construction (an item's tiles read by
        # the first sweep of a pair, written by the second: 6 n^2 b B per channel).
        sb_min = (2.0 * b * n * 8 + 16.0 * n) * nl
        ver = capi.get_option("sb2st_version")
        two_step = ver == 9 or (ver == 0 and n >= 512)

        def from_stats(sub):
            hit = [(k, v) for k, v in kstats.items() if sub in k]
            if not hit:
                return None
            calls = sum(v[1] for _, v in hit)
            return {"launches_per_step": calls / 5.0, "avg_launch_ms": sum(v[0] * v[1] for _, v in hit) / calls,
                    "kernel_ms_per_step": sum(v[0] * v[1] for _, v in hit) / 5.0, "source": stats_file,
                    "traffic_bytes_per_launch": pmc_bytes(sub), "traffic_source": pmc_file}

        if two_step:
            # Two steps (csrc/sbr2.hip).  Step 1, sb2sb_mfma_kernel: one launch per wavefront of independent chase items (sweep of
            # 16 columns s, step k, t = k + 3 s); an item reads and writes a 64 x 64 bulge tile, the lower triangle of a 64 x 64
            # diagonal tile and the next 64 x 64 tile.  Step 2, sb16st_kernel: ONE launch; every pass of 8 sweeps streams the
            # remaining band (32 rows of 8 B per column) through an LDS window once: read + write.
            items = sum(max(0, -(-(n - 16 * (s_ + 1)) // 64)) for s_ in range((n - 1) // 16))
            b1 = items * (2 * 64 * 64 * 8 + 2 * 2080 * 8 + 2 * 64 * 64 * 8) * nl
            b2 = sum(2 * 32 * 8 * (n - s0_) for s0_ in range(0, n - 2, 8)) * nl
            kern.append({"kernel": "sb2sb_mfma_kernel (band 64 -> 16, one launch per wavefront) + sb16st_kernel (band 16 -> 1, one launch)",
                         "bound": "hbm", "launch_ms": sb_ms, "launch_ms_source": "HIP events around the stage, this run",
                         "bytes_min": sb_min, "bytes_model": b1 + b2, "bytes_model_sb2sb": b1, "bytes_model_sb16st": b2,
                         "chase_items_sb2sb_per_channel": items,
                         "achieved": (b1 + b2) / (sb_ms * 1e-3) / 1e9 if sb_ms > 0 else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": (b1 + b2) / (sb_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if sb_ms > 0 else None,
                         "split_from_profile": {"sb2sb_mfma_kernel": from_stats("sb2sb_mfma_kernel"),
                                                "sb16st_kernel": from_stats("sb16st_kernel")},
                         "traffic": (sum(v["traffic_bytes_per_launch"] * v["launches"] for k_, v in pmc["kernels"].items()
                                         if "sb2sb_mfma_kernel" in k_ or "sb16st_kernel" in k_) or None) if pmc else None,
                         "traffic_source": pmc_file,
                         "note": "step 2 is bound by the serial chase (one 16 x 16 item per wave and step, ~1.8 us per step), "
                                 "not by memory; step 1 by item latency at two workgroups per CU"})
        else:
            sb_model = 6.0 * n * n * b * nl
            kern.append({"kernel": "sb2st_kernel_v7<0>", "bound": "hbm", "launch_ms": sb_ms, "launch_ms_source": "HIP events, this run",
                         "bytes_min": sb_min, "bytes_pass_model": sb_model,
                         "achieved": sb_model / (sb_ms * 1e-3) / 1e9 if sb_ms > 0 else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": sb_model / (sb_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if sb_ms > 0 else None,
                         "frac_of_min_bytes": sb_min / (sb_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if sb_ms > 0 else None,
                         "traffic": pmc_bytes("sb2st_kernel_v7"), "traffic_source": pmc_file})
        # rank-128 update executes 0.55 of 2 n^3 / 3 ... per channel: sum over panels of the valid tiles; symm 2 n^3 / 3
        for kname, flop, label in (("gemm2_kernel<128, 128", 0.55 * 4.0 / 3.0 * n ** 3 * nl, "rank-128 update (syr2k)"),
                                   ("gemm2_kernel<64, 128", 2.0 / 3.0 * n ** 3 * nl, "symm Y = A22 W")):
            hit = [(k, v) for k, v in kstats.items() if k.startswith(kname)]
            if not hit:
                continue
            tot_ms = sum(v[0] * v[1] for _, v in hit)            # all launches of all steps of the profiled run
            calls = sum(v[1] for _, v in hit)
            steps_prof = 5.0                                      # tools/gpu_profiles.sh: --steps 4 --warmup 1
            per_step_ms = tot_ms / steps_prof
            # achieved: the MFMA pipe's busy share of the kernel running ALONE (counter pass, profiles/*_mfma_util.json:
            # SQ_VALU_MFMA_BUSY_CYCLES / SIMD-cycles) x peak; in the pipeline the launches of two channel groups overlap,
            # so the sum of their durations (kernel_ms_per_step) exceeds the wall time they occupy
            util = next((v["mfma_util"] for k, v in mfma.items() if k.startswith(kname)), None)
            ach = util * FP64_PEAK_TFLOPS if util is not None else flop / (per_step_ms * 1e-3) / 1e12
            kern.append({"kernel": kname + ", ...>", "what": label, "bound": "mfma", "launches_per_step": calls / steps_prof,
                         "kernel_ms_per_step": per_step_ms, "launch_ms_source": stats_file,
                         "flop_per_step": flop, "achieved": ach, "peak": FP64_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": ach / FP64_PEAK_TFLOPS,
                         "achieved_source": (mfma_file + " (MFMA pipe busy, kernel alone)") if util is not None else "flop / sum of overlapped launch durations",
                         "tflops_from_overlapped_launch_sums": flop / (per_step_ms * 1e-3) / 1e12,
                         "traffic": pmc_bytes(kname), "traffic_source": pmc_file})
        # the path as a whole, SURVEY 8(d): F(n) flop per l-channel against the fp64 peak of the GPUs used
        ach = F * value / 1e12
        roof = {"bound": "mfma", "achieved": ach, "peak": FP64_PEAK_TFLOPS * world, "unit": "TFLOP/s",
                "frac": ach / (FP64_PEAK_TFLOPS * world),
                "scope": "whole path: F(n) = 4/3 n^3 + 4 n^2 k flop per l-channel (SURVEY 8d) x eigensolves/s, against the fp64 "
                         "matrix/vector peak of %d GPU(s)" % world,
                "traffic": (sum(v["traffic_bytes_per_launch"] * v["launches"] for v in pmc["kernels"].values()) if pmc else None),
                "traffic_note": "HBM bytes of ONE STEP (all kernels: sum of launches x bytes per launch), from the committed profile "
                                "%s -- not measured in this run" % pmc_file,
                "kernels": kern}
        out = {
            "metric": "l-channel eigensolves/sec at N_bsp=%d fp64" % n, "value": value, "unit": "eigensolves/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "Hydrogen Coulomb l=%d..%d, N_bsp=%d, k=%d, KIND_GRID=0 rb=%g (BASELINE configs[3]); "
                                   "%s" % (0, total - 1, n, args.k, args.rb,
                                           ("%d l-channels per GPU" % args.channels) if args.scaling == "weak" else
                                           ("%d l-channels in total, %s per GPU" % (total, "/".join(str(c) for c in sorted(set(counts)))))),
                       "channels_per_gpu": counts, "channels_total": total, "eigenvector_owner_rank": owner,
                       "parallelism": "l-sharded x%d (bspatom_amd/parallel.py), %s all-gather of spectra"
                                      % (world, "RCCL" if use_dist else "no (single process)")},
            "roofline": roof,
            "stage_ms_per_step_rank0": dict(zip(names + ["total_device"], [float(x) for x in stage_ms])),
            "rydberg_max_rel_err_n<=8": ryd,
        }
        if not args.no_cpu_baseline and world == 1:       # reported at N=1 only (rank 0's host cores)
            out["cpu_baseline"] = cpu_baseline(args.cpu_sample_nfun, args.k)
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    prob.close()


if __name__ == "__main__":
    main()
