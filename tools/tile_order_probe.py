#!/usr/bin/env python3
"""Gate-less fused pass (load -> LDS -> store) for a few tile-bit sets, used to A/B tile-order variants
of k_tile (probe build: QSIM_DEBUG_SKIP_GATES=4 + QSIM_DEBUG_TILE_BITS).  python tools/tile_order_probe.py [n]"""
import os
import subprocess
import sys

# probe build of the library (make -C quantum_simulations_amd/csrc probes): the product build reads no probe knobs
_PROBES = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "quantum_simulations_amd", "libqsim_hip_probes.so")
os.environ.setdefault("QSIM_LIBRARY", os.path.abspath(_PROBES))

n = int(sys.argv[1]) if len(sys.argv) > 1 else 28
child = r'''
import sys
sys.path.insert(0, ".")
import numpy as np
from quantum_simulations_amd.kernel.device import DeviceChunk
from quantum_simulations_amd.kernel import gates as gt
n = int(sys.argv[1])
dev = DeviceChunk.empty(n); dev.init_random(1)
ops = [([3], gt.H()), ([4], gt.H())]
dev.apply_ops(ops); dev.sync()
ts = []
for _ in range(7):
    dev.time_begin(); dev.apply_ops(ops); ts.append(dev.time_end())
print("%.3f" % float(np.median(ts)))
'''
SETS = [[3, 4, 5, 6, 7, 8, 9, 10], [3, 4, 5, 6, 7, 8, 11, 12], [3, 4, 5, 6, 18, 19, 20, 21], [20, 21, 22, 23, 24, 25, 26, 27],
        [5, 6, 8, 10, 15, 18, 21, 26], [3, 4, 5, 6, 18, 20, 25, 26], [15, 17, 21, 22, 23, 25, 26, 27], [4, 7, 8, 9, 10, 11, 14, 16],
        [7, 10, 14, 17, 18, 22, 24, 25], [3, 4, 5, 9, 12, 14, 19, 25]]
row = []
for bits in SETS:
    env = dict(os.environ, QSIM_DEBUG_SKIP_GATES="4", QSIM_DEBUG_TILE_BITS=",".join(map(str, bits)))
    out = subprocess.run([sys.executable, "-c", child, str(n)], env=env, capture_output=True, text=True, timeout=300)
    row.append(out.stdout.strip().splitlines()[-1] if out.returncode == 0 else "nan")
print(" ".join(row), flush=True)
