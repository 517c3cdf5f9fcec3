#!/usr/bin/env python3
"""Per-pass times (QSIM_DEBUG_STATS=2, probe build) of the bench circuit under the engine's chosen layout and under the
identity-line layout (model placement only).   python tools/layout_pass_times.py [n]"""
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
os.environ["QSIM_DEBUG_STATS"] = "2"
os.environ.setdefault("QSIM_LIBRARY", str(ROOT / "quantum_simulations_amd" / "libqsim_hip_probes.so"))
sys.path.insert(0, str(ROOT))
import numpy as np  # noqa: E402

from quantum_simulations_amd.circuit.fusion import batch_levels  # noqa: E402
from quantum_simulations_amd.circuit.io import levelize, validate_circuit_dict  # noqa: E402
from quantum_simulations_amd.circuits import random_1q_cx_circuit  # noqa: E402
from quantum_simulations_amd.kernel.device import DeviceChunk, pack_ops  # noqa: E402
from quantum_simulations_amd.runner import engine as eng_mod  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 28
cd = validate_circuit_dict(random_1q_cx_circuit(n, depth=40))
batches = [p["local_ops"] for p in batch_levels(levelize(cd), n)]
state = DeviceChunk.empty(n)
for name, k in (("line qubits searched", 128), ("line qubits 0 1 2", 0)):
    l2p, masks, info = eng_mod.choose_plan_layout(n, batches, k)
    packed = [pack_ops([([l2p[q] for q in qs], U) for qs, U in ops]) for ops in batches]
    print(f"==== {name}: {info}", file=sys.stderr, flush=True)
    for rep in range(2):
        state.init_zero(True)
        if rep:
            print("---- second execution", file=sys.stderr, flush=True)
        for p, m in zip(packed, masks):
            state.apply_ops_tiled(p, m)
        state.sync()
state.close()
