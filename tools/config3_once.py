#!/usr/bin/env python3
"""BASELINE config 3 once, for rocprofv3 (tools/profile_round.sh <tag> c3): H on every target of a 30-qubit random state, T
and CNOT on a few targets, and the dense blocks k = 3 .. 6 -- one launch per gate, three repeats each -- so that the
per-gate kernels (k_gate<2>, k_gate_shuffle<1,1>, k_gate<1>, k_dense_mfma2<K>) have a kernel-trace and a PMC summary taken
with the CURRENT kernel sources next to the fused pass's (the per-gate evidence dated from round 1).
    python tools/config3_once.py [n]"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from quantum_simulations_amd.kernel import gates as gt  # noqa: E402
from quantum_simulations_amd.kernel.device import DeviceChunk  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = DeviceChunk.empty(n)
dev.init_random(30)
H, T, CX = gt.H(), gt.T(), gt.CNOT()
rng = np.random.default_rng(4)
for _ in range(3):
    for q in range(n):
        dev.apply_1q(q, H)
    for q in (0, 3, 7, 12, 20, n - 1):
        dev.apply_1q(q, T)
        if q + 1 < n:
            dev.apply_2q(q, q + 1, CX)
    for k in (3, 4, 5, 6):
        M = np.linalg.qr(rng.standard_normal((1 << k, 1 << k)) + 1j * rng.standard_normal((1 << k, 1 << k)))[0]
        for qs in (list(range(3, 3 + k)), list(range(n - k, n)), list(range(k))):
            dev.apply_fused_k(qs, M)
dev.sync()
print("norm2", dev.norm2())
dev.close()
