#!/usr/bin/env python3
"""Per-descriptor cost of the fused tile pass: ONE pass holding M copies of one gate kind (on tile
qubit 5, controls on qubit 9), time vs M -> fixed cost of a pass and slope per gate kind.
    python tools/gate_cost_probe.py [n_qubits]"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from quantum_simulations_amd.kernel import gates as gt  # noqa: E402
from quantum_simulations_amd.kernel.device import DeviceChunk  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 28
dev = DeviceChunk.empty(n)
dev.init_random(3)
kinds = {"Z": ([5], gt.Z()), "T": ([5], gt.T()), "X": ([5], gt.X()), "H": ([5], gt.H()),
         "RY": ([5], gt.RY(0.3)), "CNOT(9,5)": ([9, 5], gt.CNOT()), "CNOT(20,5)": ([20, 5], gt.CNOT()),
         "CZ(9,5)": ([9, 5], gt.CZ())}
for name, op in kinds.items():
    row = []
    for m in (4, 16, 32, 64, 120):
        ops = [op] * m
        passes = dev.apply_ops(ops)
        dev.sync()
        ts = []
        for _ in range(5):
            dev.time_begin()
            dev.apply_ops(ops)
            ts.append(dev.time_end())
        row.append((m, passes, float(np.median(ts))))
    slope = (row[-1][2] - row[1][2]) / (row[-1][0] - row[1][0])
    print(f"{name:12s} " + "  ".join(f"M={m}: {t:.3f} ms ({p}p)" for m, p, t in row) + f"   slope {slope * 1e3:.1f} us/gate", flush=True)
