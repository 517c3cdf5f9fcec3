"""`python3 bench.py --gpus 4 --rehearsal` on ONE MI355X through the launcher-free path: the parent starts four ranks that
share the device (real HIP kernels on real shards, host-staged gloo exchange), relays ONE JSON line that says so, and the
line carries the amplitude checks of the partitioned random circuits against a one-GPU run."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def test_four_ranks_sharing_one_gpu_through_the_launcher():
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE") and not k.startswith("QSIM_")}
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "4", "--rehearsal", "--local-qubits", "20",
                          "--steps", "2", "--warmup", "1"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-5000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    doc = json.loads(lines[0])
    assert doc["n_gpus"] == 4 and doc["exchange"] == "gloo-rehearsal" and doc["exchange_api"] == "torch"
    assert doc["config"]["n_qubits"] == 22 and "invalid" not in doc
    cfg = doc["baseline_configs"]
    assert all(r["pass_1e-10"] for r in cfg["config5"])
    assert all(r["max_abs_err_sampled_host_check"] < 1e-10 for r in cfg["config5"])      # ("plus sampled host check")
    for key, labels in (("config4", ("staged", "unstaged")), ("random_1q_cx", ("staged",))):
        for label in labels:
            assert cfg[key][label]["fingerprint_max_abs_diff_vs_single_gpu"] < 1e-10
    assert doc["roofline"]["kernel"].startswith("k_tile") and doc["value"] > 0
    # one N > 1 run answers the open multi-GPU questions -- measured re-layouts by m (null rates where only
    # RCCL events can tell), the step with unfused re-layouts, the other exchange API (skipped in rehearsal: says why), and
    # the wall-clock sections
    rl = doc["relayout_measured"]
    assert [r["m"] for r in rl] == [1, 2] and all(r["wall_ms_pack_exchange_unpack"] > 0 and r["bytes_sent_per_rank"] > 0 for r in rl)
    assert all(r["exchange_event_ms"] is None and r["modelled_pass_units"] in (10.4, 5.2) for r in rl)      # (gloo rehearsal: no device events)
    ab = doc["fused_relayout_ab"]
    assert ab["ms_per_step_fused"] > 0 and ab["ms_per_step_unfused"] > 0 and ab["hbm_passes_unfused"] >= ab["hbm_passes_fused"]
    assert doc["other_exchange_api"]["exchange_api"] == "cabi" and doc["other_exchange_api"]["ran"] is False and "rehearsal" in doc["other_exchange_api"]["skipped"]
    wc = doc["wall_clock"]
    assert wc["skipped"] == [] and wc["total_s"] > 0 and [s_["name"] for s_ in wc["sections"]][:2] == [
        "plan (start layout search, stage boundaries + tile passes of every execution)", "warm-up + the timed steps"]
    assert doc["relayout_pipeline"] is True and doc["plan_seconds"] > 0
    # ... and with no budget at all every optional section is skipped and named, the line stays valid
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "4", "--rehearsal", "--local-qubits", "20", "--steps", "1",
                          "--warmup", "1", "--budget-seconds", "0", "--no-configs", "--no-relayout-pipeline"], cwd=ROOT, env=env,
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-5000:]
    doc = json.loads([ln for ln in out.stdout.splitlines() if ln.strip()][0])
    assert doc["relayout_measured"] is None and doc["fused_relayout_ab"] is None and doc["single_gpu_same_local_size"] is None
    assert len(doc["wall_clock"]["skipped"]) == 3 and doc["relayout_pipeline"] is False and "invalid" not in doc
    # ... and a second-exchange-API section that never returns (test hook) does not lose the line: the watchdog prints it
    # with the section marked as abandoned and every rank exits 0
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "4", "--rehearsal", "--local-qubits", "20", "--steps", "1",
                          "--warmup", "1", "--no-configs", "--ab-steps", "1", "--other-api-timeout", "5"], cwd=ROOT,
                         env=dict(env, BENCH_TEST_OTHER_API_HANG="1"), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-5000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1
    doc = json.loads(lines[0])
    assert "abandoned" in doc["other_exchange_api"]["error"] and doc["value"] > 0 and "invalid" not in doc


def test_single_gpu_line_carries_the_contract_fields():
    """`python3 bench.py` at N = 1 (a small state, short CPU column): ONE JSON line with the fields the driver reads --
    metric / value / unit / n_gpus / steps / warmup / ms_per_step / higher_is_better / scaling / vs_baseline / dtype /
    data / config.workload, `roofline` {bound, achieved, peak, unit, frac, traffic}, `cpu_baseline` {value, unit, cores,
    kind, sample} -- exit code 0, no `invalid`; and exit code 2 with a QSIM_* knob in the environment."""
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE") and not k.startswith("QSIM_")}
    cmd = [sys.executable, str(ROOT / "bench.py"), "--local-qubits", "22", "--steps", "3", "--warmup", "1", "--sweep-qubits", "20",
           "--fused-qubits", "0", "--sustain-seconds", "0", "--cpu-seconds", "2"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["dtype"] == "f64" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert set(d["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"} and d["roofline"]["bound"] == "hbm"
    assert abs(d["roofline"]["frac"] - d["roofline"]["achieved"] / d["roofline"]["peak"]) < 1e-3
    assert set(d["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample"} and d["cpu_baseline"]["kind"] == "port"
    assert d["parity_max_abs_diff_vs_cpu_prefix"] < 1e-10 and "invalid" not in d
    sweep = d["sweep30"]                         # (config 3's section, here at 20 qubits: one launch per gate and target)
    assert set(sweep["rows"]) == {"H(q)", "T(q)", "CNOT(q,q+1)", "CNOT(0,q)"} and len(sweep["rows"]["H(q)"]["frac_of_8TBps"]) == 20
    for k in (3, 4, 5, 6):                       # ... and the dense blocks on the matrix cores
        blocks = sweep["dense_blocks"][f"k={k}"]
        assert len(blocks["qubit_sets"]) == 5 and all(f > 0 for f in blocks["frac_of_8TBps"]) and "mfma" in blocks["kernel"]
        assert blocks["bound"] == ("mfma" if k == 6 else "hbm") and all(x > 0 for x in blocks["achieved_TFLOPs"])
    assert abs(sweep["norm2_after"] - 1.0) < 1e-10
    # what ONE call of the drop-in entry points costs, next to the steady-state value; item 7: the
    # streaming ceiling seen in this run and the layout facts at the top level; ADVICE r04: the timed plan itself is checked
    api = d["api_path"]
    for entry in ("single_node.run", "Driver.run_circuit"):
        assert api[entry]["seconds"] > 0 and api[entry]["gate_apps_per_s"] > 0 and api[entry]["fingerprint_abs_diff_vs_engine"] < 1e-10
    assert d["first_execution_ms"] > 0 and d["plan_seconds"] >= 0 and d["layout"] in ("identity", "searched")
    assert "qubit_layout_seconds" in d and "value_is" in d["config"]
    sc = d["stream_ceiling"]
    assert sc["GBps"] == max(sc["candidates_GBps"].values()) and sc["GBps"] >= d["copy_ceiling"]["GBps"] * 0.98
    assert abs(d["roofline"]["frac_of_achievable"] - d["roofline"]["achieved"] / sc["GBps"]) < 1e-3
    assert d["timed_plan_check"]["fingerprint_abs_diff_vs_identity_layout_plan"] < 1e-10
    bad = subprocess.run(cmd, cwd=ROOT, env=dict(env, QSIM_PLAN_LOOKAHEAD="0"), capture_output=True, text=True, timeout=120)
    assert bad.returncode == 2 and "refusing" in bad.stdout
