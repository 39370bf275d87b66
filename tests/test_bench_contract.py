"""The driver's contract with bench.py and __graft_entry__: one JSON line with the agreed fields on a GPU, a loud
failure without one."""
import json
import os
import subprocess
import sys

import pytest

from zignal_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*args, timeout=300):
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(args), cwd=ROOT, capture_output=True, text=True,
                          timeout=timeout)


def test_bench_and_smoke_fail_loudly_without_a_gpu():
    if _lib.device_count() > 0:
        pytest.skip("a GPU is present")
    r = run_bench("--steps", "1", "--warmup", "0")
    assert r.returncode != 0 and r.stdout.strip() == "" and "no GPU" in r.stderr
    code = "import __graft_entry__ as g; g.smoke()"
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0


REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline"}


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [(), ("--dtype", "f32"), ("--layout", "capture"), ("--mode", "period", "--cpu-seconds", "0")])
def test_bench_prints_one_json_line_with_the_agreed_fields(extra):
    r = run_bench("--order", "14", "--channels", "32", "--periods", "3", "--steps", "3", "--warmup", "1", "--cpu-seconds", "1", *extra)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert REQUIRED <= set(d)
    assert d["unit"] == "Gsamples/s" and d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert d["dtype"] == ("f32" if "f32" in extra else "f64")
    assert "workload" in d["config"] and "model" not in d["config"]
    L = (1 << 14) - 1
    assert abs(d["value"] - 32 * 3 * L / (d["ms_per_step"] * 1e-3) / 1e9) <= 1e-6 * d["value"]
    roof = d["roofline"]
    assert roof["bound"] == "hbm" and roof["unit"] == "GB/s" and roof["peak"] == 8000.0
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-12 and roof["achieved"] > 0
    assert d["system_check_max_abs_err"] < (1e-4 if "f32" not in extra else 1e-3)
    assert "capture_placement" in d and "configs" not in d          # small captures are not searched; extra configs ride on the default workload only
    assert "frac_of_measured_copy_rate" not in roof
    if "period" in extra:
        assert d["cpu_baseline"] is None
    else:
        cpu = d["cpu_baseline"]
        assert cpu["kind"] == "port" and cpu["cores"] == 1 and cpu["value"] > 0 and cpu["unit"] == "Gsamples/s" and cpu["sample"]
        assert d["parity_max_rel_err_vs_oracle"] <= (1e-12 if "f32" not in extra else 1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("launcher", ["bench.py --gpus 2", "torch.distributed.run"])
def test_two_ranks_print_one_json_line_too(launcher):
    """N > 1, both ways it is launched: `bench.py --gpus 2` starting its own two ranks, and the driver's
    torch.distributed.run (which only sets the environment: the ranks themselves never import torch). Both ranks share
    the card on a one-GPU box. Rank 0 alone prints, and nothing else (RCCL's banner) reaches stdout."""
    args = ["--gpus", "2", "--order", "14", "--channels", "32", "--periods", "3", "--steps", "3", "--warmup", "1"]
    if launcher.startswith("bench.py"):
        env = dict(os.environ, MASTER_PORT="29543")
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
            env.pop(k, None)
        cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + args
    else:
        env = dict(os.environ)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", "29541", os.path.join(ROOT, "bench.py")] + args
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert REQUIRED <= set(d) and d["n_gpus"] == 2 and d["scaling"] == "weak" and d["cpu_baseline"] is None
    L = (1 << 14) - 1
    assert abs(d["value"] - 2 * 32 * 3 * L / (d["ms_per_step"] * 1e-3) / 1e9) <= 1e-6 * d["value"]
    # every rank's own step time is in the line; the job's is the slowest rank's
    pr = d["per_rank"]
    assert len(pr["ms_per_step"]) == 2 and abs(max(pr["ms_per_step"]) - d["ms_per_step"]) <= 1e-9 * d["ms_per_step"] and pr["min_ms"] <= pr["max_ms"]
    # the RCCL leg rides on every N > 1 run. Two ranks on two GPUs: a two-rank communicator moved the partial sums.
    # Two ranks sharing the one GPU of a test box: RCCL refuses the communicator, every rank learns of it, and the
    # line says so instead of a number. Either way the line names the librccl that answered: the ROCm installation's,
    # not one that came along with a Python package.
    pm = d["period_mode"]
    assert pm["librccl_path"].startswith("/opt/rocm") and "torch" not in pm["librccl_path"] and pm["librccl_version"] > 0, pm
    if "error" in pm:
        assert _lib.device_count() < 2, pm["error"]
        assert "ncclCommInitRank" in pm["error"]
    else:
        assert pm["rccl_ranks"] == 2 and pm["ms_per_step"] > 0 and pm["GBps_exchanged_per_gpu"] > 0
        assert pm["system_check_max_abs_err"] < 1e-4 and len(pm["ms_per_step_per_rank"]["all"]) == 2


@pytest.mark.gpu
def test_a_stuck_period_leg_prints_the_line_and_leaves_with_a_non_zero_status():
    """A leg that does not finish must not cost the headline its line, and must not look like success either: with a
    deadline it cannot meet (0 s) the line is printed with period_mode.error and the status is 4."""
    env = dict(os.environ, MASTER_PORT="29545")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--order", "14", "--channels", "32", "--periods", "3", "--steps", "2",
                        "--warmup", "1", "--period-leg-timeout", "0"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-2000:] + r.stderr[-2000:]
    d = json.loads(lines[0])
    assert d["value"] > 0 and "did not finish" in d["period_mode"]["error"]
    assert r.returncode == 4, r.returncode


@pytest.mark.gpu
def test_extra_config_entry_carries_roofline_parity_and_cpu_baseline():
    """bench.py's `configs` entries (BASELINE configs 1, 2, 4, 5 at full size in the driver's run), on a small stand-in."""
    sys.path.insert(0, ROOT)
    import bench
    keep = {}
    e64 = bench.run_extra_config(("config5_f64", "stand-in", 13, 32, 1, "f64", 3, 2, 1), 0, 1.0, keep)
    e32 = bench.run_extra_config(("config5_f32", "stand-in", 13, 32, 1, "f32", 3, 2, 0), 0, 1.0, keep)
    e1 = bench.run_extra_config(("config1", "stand-in", 12, 1, 1, "f64", 3, 1, 1), 0, 1.0, keep)
    for e, tol in ((e64, 1e-12), (e32, 1e-5), (e1, 1e-12)):
        assert {"workload", "dtype", "ms_per_step", "value", "roofline", "parity_max_rel_err_vs_oracle", "cpu_baseline",
                "system_check_max_abs_err"} <= set(e)
        r = e["roofline"]
        assert r["alg_bytes_per_step"] == 2 * ((1 << e["order"]) - 1) * e["channels"] * (8 if e["dtype"] == "f64" else 4)
        assert abs(r["frac"] - r["achieved"] / 8000.0) < 1e-12 and e["ms_per_step"] > 0
        assert e["parity_max_rel_err_vs_oracle"] <= tol
    assert e64["cpu_baseline"]["kind"] == "port" and e64["cpu_baseline"]["value"] > 0
    assert e32["cpu_baseline"] is None and 0 < e32["fp32_vs_fp64_rel_err"] <= 1e-5 and e32["fp32_speedup_over_fp64"] > 0
    assert e1["host_api_ms_per_call"] > 0 and e1["cpu_baseline"]["cores"] == 1
