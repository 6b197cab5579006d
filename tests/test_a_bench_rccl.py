"""bench.py through the driver's multi-GPU launch line at N = 1: `python -m torch.distributed.run --nproc-per-node 1 ...`
initialises RCCL (backend "nccl") and sends the spectra through `all_gather_into_tensor` (bspatom_amd/parallel.py issues
the collective at world size 1 too; `config.collective_calls` in the line counts the calls), so the collective branch of the
bench runs on hardware once per test session.  What has NOT run on hardware in this repo: any world size above 1 (the
driver owns the 8-GPU node; tests/test_host_cpu.py rehearses N = 2 over gloo).  The file sorts first: the child process
starts before this pytest process has touched the GPU."""
import json
import os
import subprocess
import sys
import pytest
from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("scaling,route", [("weak", 0), ("strong", 0), ("weak", 1)])
def test_bench_under_torchrun_one_rank(scaling, route):
    port = 29600 + os.getpid() % 300 + (0 if scaling == "weak" else 1) + 2 * route
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "1",
           "--nfun", "1024", "--rb", "200", "--channels", "8", "--scaling", scaling, "--no-cpu-baseline", "--route", str(route)]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    lines = [l for l in p.stdout.split("\n") if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["scaling"] == scaling and d["unit"] == "eigensolves/s" and d["value"] > 0
    assert d["config"]["channels_total"] == 8 and d["config"]["channels_per_gpu"] == [8]
    assert "RCCL" in d["config"]["parallelism"] and d["config"]["launched_by"] == "torch.distributed.run"
    # the spectra really went through all_gather_into_tensor on RCCL: once per step and warm-up step (world size 1 no
    # longer returns early, bspatom_amd/parallel.py)
    assert d["config"]["collective_calls"] == 2
    assert d["rydberg_max_rel_err_n<=8"] < 1e-4        # n = 8 reaches the wall of the rb = 200 box (1.5e-5); n <= 4: 1e-10
    r = d["roofline"]
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    # per-kernel entries: launch durations measured in this run (HIP events around every launch), frac = work / time
    ks = {k["kernel"].split("(")[0].split("<")[0].strip(): k for k in r["kernels"]}
    # default route at k = 9: the band route (csrc/crawford.hip); --route 1: the dense route, whose line lists the two big GEMMs
    assert d["route"] == ("dense" if route == 1 else "band")
    want = ("gemm2_kernel",) if route == 1 else ("crawford_item4_kernel", "sbr_rows_kernel", "bisect3_kernel")
    for name in want:
        assert name in ks, list(ks)
    for k in r["kernels"]:
        if "launches_per_step" in k:
            assert k["launches_per_step"] > 0 and k["kernel_ms_per_step"] > 0 and "this run" in k["launch_ms_source"]
            assert 0 < k["frac"] < 1 and abs(k["frac"] - k["achieved"] / k["peak"]) < 1e-9
    if route == 0:
        # the two extra legs of a one-GPU run: north_star's dense route and the all-eigenvectors call, measured in the same run
        assert d["dense_two_stage"]["value"] > 0 and 0 < d["dense_two_stage"]["roofline"]["frac"] < 1
        assert d["full_V"]["unit"] == "channels/s" and d["full_V"]["check_64_vectors"]["max_S_orthonormality_defect"] < 1e-11
        assert d["roofline"]["dense_algorithm_ceiling"]["eigensolves_per_s_at_100_percent_of_fp64_peak"] > 0


def test_sharded_host_under_torchrun(tmp_path):
    """`python -m torch.distributed.run --nproc-per-node 1 -m bspatom_amd.host` (the multi-GPU KIND_PI = 0 host at N = 1:
    RCCL initialised, spectra through parallel.gather_spectra): Enl.dat, wf_n0.dat and stdout byte for byte what the
    single-process Python host writes for the same input."""
    import numpy as np
    from conftest import golden_input
    inp = golden_input("lin256")
    out1 = tmp_path / "sharded"; out1.mkdir()
    port = 29950 + os.getpid() % 40
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), "-m", "bspatom_amd.host"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", BSPATOM_INPUT=inp, BSPATOM_OUTDIR=str(out1), PYTHONPATH=ROOT)
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    from bspatom_amd import host
    out2 = tmp_path / "single"; out2.mkdir()
    E, c, text = host.run(open(inp).read(), outdir=str(out2))
    assert open(out1 / "Enl.dat").read() == open(out2 / "Enl.dat").read()
    assert open(out1 / "wf_n0.dat").read() == open(out2 / "wf_n0.dat").read()
    assert p.stdout.rstrip("\n") == text.rstrip("\n")
