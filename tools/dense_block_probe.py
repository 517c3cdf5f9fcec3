#!/usr/bin/env python3
"""HBM rate of the dense k-qubit block kernel (qsim_apply_fused_k) at n qubits: k = 3, 4 on low / middle / high / mixed
index bits, median of 5 (HIP events), as a fraction of the 8 TB/s peak of the 32 B x 2^n a launch moves.
    python tools/dense_block_probe.py [n]"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from quantum_simulations_amd.kernel.device import DeviceChunk  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(1)
dev = DeviceChunk.empty(n)
dev.init_random(7)
for qs in ([3, 4, 5], [10, 11, 12], [n - 3, n - 2, n - 1], [5, 14, n - 2], [0, 1, 2], [3, 4, 5, 6], [9, 13, 17, 21], [n - 4, n - 3, n - 2, n - 1],
           [4, 12, 20, n - 1], [0, 1, 2, 3]):
    k = len(qs)
    M = np.linalg.qr(rng.standard_normal((1 << k, 1 << k)) + 1j * rng.standard_normal((1 << k, 1 << k)))[0]
    dev.apply_fused_k(qs, M)
    dev.sync()
    ts = []
    for _ in range(5):
        dev.time_begin()
        dev.apply_fused_k(qs, M)
        ts.append(dev.time_end())
    ms = float(np.median(ts))
    print(f"k={k} qubits {qs}: {ms:.3f} ms  frac {32 * 2 ** n / (ms * 1e-3) / 8e12:.3f}", flush=True)
print("norm2", dev.norm2())
dev.close()
