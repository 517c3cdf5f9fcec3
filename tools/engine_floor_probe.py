#!/usr/bin/env python3
"""What does a fused pass WITH ~47 gates cost when its tile has a given memory pattern?  A random 1q+CX circuit on
11 chosen qubits (the three line bits + 8 high bits) plans into single passes whose tile is exactly that set, so the
gate engine's cost can be read on the fastest pattern (3-6,12-15), the contiguous tile and slow patterns, next to the
gate-less pass over the same tile bits (second process run with QSIM_DEBUG_SKIP_GATES=1).  Probe build.
    python3 tools/engine_floor_probe.py [n_qubits]"""
import os
import sys
from pathlib import Path

os.environ.setdefault("QSIM_LIBRARY", str(Path(__file__).resolve().parent.parent / "quantum_simulations_amd" / "libqsim_hip_probes.so"))
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np  # noqa: E402

from quantum_simulations_amd.circuits import random_1q_cx_circuit  # noqa: E402
from quantum_simulations_amd.kernel import gates as gt  # noqa: E402
from quantum_simulations_amd.kernel.device import DeviceChunk  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 28
sets = {"P* 3-6,12-15": [3, 4, 5, 6, 12, 13, 14, 15], "contiguous 3-10": list(range(3, 11)),
        "3-6,16-19": [3, 4, 5, 6, 16, 17, 18, 19], "3-6,20-23": [3, 4, 5, 6, 20, 21, 22, 23],
        "bench pass 4": [4, 6, 7, 12, 14, 18, 21, 25], "bench pass 0": [4, 7, 8, 9, 10, 11, 14, 16]}
dev = DeviceChunk.empty(n)
dev.init_random(1)
for name, high in sets.items():
    if max(high) >= n:
        continue
    qubits = [0, 1, 2] + high
    for n_layers, label in ((6, "~50 gates"), (3, "~25 gates"), (12, "~100 gates: two passes")):
        small = random_1q_cx_circuit(11, depth=n_layers, seed=7)
        ops = []
        for g in small["gates"]:
            ops.append(([qubits[q] for q in g["qubits"]], gt.gate_matrix(g["gate"], g.get("params", {}))))
        passes = dev.apply_ops(ops)
        dev.sync()
        ts = []
        for _ in range(9):
            dev.time_begin()
            dev.apply_ops(ops)
            ts.append(dev.time_end())
        print(f"{name:18s} {label:24s} {len(ops):3d} gates {passes} pass(es)  {np.median(ts) / passes:.4f} ms/pass", flush=True)
