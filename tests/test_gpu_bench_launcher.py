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
    for key, labels in (("config4", ("staged", "unstaged")), ("random_1q_cx", ("staged",))):
        for label in labels:
            assert cfg[key][label]["fingerprint_max_abs_diff_vs_single_gpu"] < 1e-10
    assert doc["roofline"]["kernel"].startswith("k_tile") and doc["value"] > 0
