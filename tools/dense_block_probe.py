#!/usr/bin/env python3
"""Rate of the dense k-qubit block kernel (qsim_apply_fused_k) at n qubits: k = 3 .. 6 on low / middle / high / mixed /
line index bits, median of 5 (HIP events), as a fraction of the 8 TB/s peak of the 32 B x 2^n a launch moves and as
Tflop/s of its 8 * 2^k flop per amplitude against the 78.6 Tflop/s fp64 matrix peak.  With the probe build and
QSIM_DENSE_FORM=0 the round-4 kernels run (k <= 4): the A/B partner.  DENSE_KS="3,4" restricts the block sizes, DENSE_MORE_SETS=1
adds fifteen more positions.
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
import os
old_form = os.environ.get("QSIM_DENSE_FORM") == "0"
sets = [[3, 4, 5], [10, 11, 12], [n - 3, n - 2, n - 1], [5, 14, n - 2], [0, 1, 2], [3, 4, 5, 6], [9, 13, 17, 21], [n - 4, n - 3, n - 2, n - 1],
        [4, 12, 20, n - 1], [0, 1, 2, 3]]
if not old_form:
    sets += [[3, 4, 5, 6, 7], [10, 11, 12, 13, 14], list(range(n - 5, n)), [5, 14, n - 2, n - 9, 8], [0, 1, 2, 3, 4], [0, 1, 2, 12, n - 1],
             [3, 4, 5, 6, 7, 8], [10, 11, 12, 13, 14, 15], list(range(n - 6, n)), [5, 14, n - 2, n - 9, 8, 11], [0, 1, 2, 3, 4, 5], [0, 1, 2, 12, 20, n - 1]]
if os.environ.get("DENSE_MORE_SETS"):          # more positions, to see what a launch-shape choice generalises to
    sets += [[6, 7, 8], [15, 16, 17], [20, 21, 22], [24, 25, 26], [3, 15, 27], [1, 9, 19], [7, 8, 9, 10], [14, 15, 16, 17], [20, 21, 22, 23],
             [23, 24, 25, 26], [2, 10, 18, 26], [6, 7, 8, 9, 10], [15, 16, 17, 18, 19], [20, 21, 22, 23, 24], [2, 9, 16, 23, n - 1]]
ks = [int(x) for x in os.environ.get("DENSE_KS", "3,4,5,6").split(",")]
for qs in sets:
    k = len(qs)
    if k not in ks:
        continue
    M = np.linalg.qr(rng.standard_normal((1 << k, 1 << k)) + 1j * rng.standard_normal((1 << k, 1 << k)))[0]
    dev.apply_fused_k(qs, M)
    dev.sync()
    ts = []
    for _ in range(5):
        dev.time_begin()
        dev.apply_fused_k(qs, M)
        ts.append(dev.time_end())
    ms = float(np.median(ts))
    tf = 8.0 * (1 << k) * 2 ** n / (ms * 1e-3) / 1e12
    print(f"k={k} qubits {qs}: {ms:.3f} ms  frac of HBM peak {32 * 2 ** n / (ms * 1e-3) / 8e12:.3f}  {tf:.1f} Tflop/s = {tf / 78.6:.3f} of the fp64 matrix peak", flush=True)
print("norm2", dev.norm2())
dev.close()
