"""`python3 bench.py --gpus N` without a launcher: the process starts its own N ranks before any GPU
call, relays rank 0's single JSON line and exits non-zero -- ending the other ranks -- when one of them fails.  Here on
the CPU through `--dry-run` (the real DistributedEngine over DryBackend at full problem size, gloo)."""
import json
import os
import subprocess
import sys
import time
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def _clean_env(**extra) -> dict:
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE") and not k.startswith("QSIM_")}
    env.update(extra)
    return env


@pytest.mark.parametrize("world", [2, 4, 8])
def test_bench_starts_its_own_ranks_for_a_dry_run(world):
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", str(world), "--dry-run"], cwd=ROOT,
                         env=_clean_env(), capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    doc = json.loads(lines[0])
    assert doc["dry_run"] is True and doc["ok"] is True and doc["n_gpus"] == world
    assert doc["n_qubits"] == 30 + world.bit_length() - 1 and len(doc["workloads"]) == 5
    for w in doc["workloads"]:
        assert w["ok"] and len(w["per_rank"]) == world


def test_launcher_form_still_works_under_torch_distributed_run():
    from tests.test_distributed_gloo import _free_port
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), str(ROOT / "bench.py"), "--gpus", "2", "--dry-run"],
                         cwd=ROOT, env=_clean_env(), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    docs = [json.loads(ln) for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(docs) == 1 and docs[0]["ok"] and docs[0]["n_gpus"] == 2


def test_a_failing_rank_ends_the_run_with_a_non_zero_exit_and_no_hang():
    """Rank 1 dies at start (test hook): rank 0 would wait in the rendezvous for ever; the parent must end it."""
    t0 = time.time()
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--dry-run"], cwd=ROOT,
                         env=_clean_env(BENCH_TEST_FAIL_RANK="1"), capture_output=True, text=True, timeout=300)
    assert out.returncode == 3, (out.returncode, out.stderr[-2000:])
    assert time.time() - t0 < 120
    assert "rank 1 exited with code 3" in out.stderr
    assert not [ln for ln in out.stdout.splitlines() if ln.startswith("{")]


def test_the_parent_makes_no_gpu_call():
    """The launcher path may not import torch or load the HIP library in the parent: a process that has initialised the
    GPU must not be the one that starts the ranks (and on a GPU box would hold a context of its own on device 0)."""
    code = ("import sys; sys.argv = ['bench.py', '--gpus', '2', '--dry-run']\n"
            "import bench, subprocess\n"
            "real = subprocess.Popen\n"
            "def spy(*a, **k):\n"
            "    assert 'torch' not in sys.modules and 'quantum_simulations_amd._lib' not in sys.modules, 'GPU-side module loaded in the parent'\n"
            "    return real(*a, **k)\n"
            "subprocess.Popen = spy\n"
            "try:\n"
            "    bench.main()\n"
            "except SystemExit as e:\n"
            "    assert e.code == 0, e.code\n"
            "assert 'torch' not in sys.modules and 'quantum_simulations_amd._lib' not in sys.modules\n"
            "print('PARENT-CLEAN')\n")
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=_clean_env(), capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "PARENT-CLEAN" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


def test_gpu_count_must_be_a_power_of_two():
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "3", "--dry-run"], cwd=ROOT, env=_clean_env(),
                         capture_output=True, text=True, timeout=120)
    assert out.returncode != 0 and "power of two" in out.stderr


def test_wall_clock_budget_skips_optional_sections_and_says_so():
    """an N > 1 run prints every section's start to stderr and skips the OPTIONAL ones -- naming them
    in the line -- once the budget is spent; every rank must take the same decision (`agree` = a max over the ranks)."""
    sys.path.insert(0, str(ROOT))
    import bench
    now = [1000.0]
    logged, agreed = [], []

    def agree(x):
        agreed.append(x)
        return x + 5.0                      # (another rank is five seconds further on: its clock decides)
    b = bench.SectionBudget(400.0, t0=1000.0, clock=lambda: now[0], agree=agree, log=logged.append)
    assert b.begin("plan")
    now[0] += 30
    assert b.begin("timed steps")
    now[0] += 300
    assert b.begin("re-layout measurements", optional=True)          # 330 + 5 s < 400
    now[0] += 66
    assert not b.begin("fused on / off", optional=True)              # 396 + 5 s > 400: skipped ...
    assert b.begin("configs")                                         # ... a mandatory section still runs
    now[0] += 50
    assert not b.begin("other exchange API", optional=True)
    calls = len(agreed)
    rep = b.report()
    assert len(agreed) == calls, "report() runs on rank 0 alone: it must not enter a collective"
    assert rep["skipped"] == ["fused on / off", "other exchange API"] and rep["limit_s"] == 400.0
    assert [s_["name"] for s_ in rep["sections"]] == ["plan", "timed steps", "re-layout measurements", "configs"]
    assert [s_["seconds"] for s_ in rep["sections"]] == [30.0, 300.0, 66.0, 50.0]
    assert len(agreed) >= 6 and any("SKIPPED fused on / off" in ln for ln in logged) and any("timed steps" in ln for ln in logged)


def test_dry_run_line_names_the_fields_a_device_run_fills_in():
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--dry-run"], cwd=ROOT,
                         env=_clean_env(), capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    doc = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
    for key in ("relayout_measured", "fused_relayout_ab", "other_exchange_api", "wall_clock"):
        assert key in doc and doc[key] is None
    # ... and a projection that says of itself that it is one: compute from the measured one-GPU pass time, the exchange
    # from the planner's link model with nothing hidden behind compute
    for w in doc["workloads"]:
        proj = w["projection_model_not_measured"]
        assert proj["step_ms"] == pytest.approx(proj["compute_ms_per_execution"] + proj["exchange_ms_per_execution_upper_bound"], abs=0.2)
        assert proj["gate_apps_per_s"] > 0
