#!/usr/bin/env python3
"""Memory-pattern probe of the fused tile pass: time of a gate-less pass (load -> LDS -> store) at
n qubits as a function of WHICH 8 high index bits the tile takes (QSIM_DEBUG_SKIP_GATES=4 +
QSIM_DEBUG_TILE_BITS; one child process per configuration because the knobs are read at load).
    python tools/tile_bits_probe.py [n_qubits]"""
import itertools
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
for _ in range(5):
    dev.time_begin(); dev.apply_ops(ops); ts.append(dev.time_end())
print("%.3f" % float(np.median(ts)))
'''


def measure(bits):
    env = dict(os.environ, QSIM_DEBUG_SKIP_GATES="4", QSIM_DEBUG_TILE_BITS=",".join(map(str, bits)))
    out = subprocess.run([sys.executable, "-c", child, str(n)], env=env, capture_output=True, text=True, timeout=300)
    return float(out.stdout.strip().splitlines()[-1]) if out.returncode == 0 else float("nan")


base = [3, 4, 5, 6, 7, 8, 9]
if len(sys.argv) > 2:        # custom sets: "20,21,24;20,21,25;..." (padded with the low bits 3, 4, ...)
    for spec in sys.argv[2].split(";"):
        hs = [int(x) for x in spec.split(",")]
        bits = base[:max(0, 8 - len(hs))] + hs          # (nine bits for a 12-bit-tile build: given in full)
        print("set", hs, measure(bits), flush=True)
    sys.exit(0)
print("contiguous", base + [10], measure(base + [10]), flush=True)
for b in range(11, n):
    print("single", b, measure(base + [b]), flush=True)
pairs = list(itertools.combinations([18, 20, 21, 24, 25], 2)) + [(22, 23), (26, 27), (22, 26), (23, 27), (19, 23)]
for a, b in pairs:
    print("pair", a, b, measure(base[:6] + [a, b]), flush=True)
for hs in ([20, 21, 24, 25], [22, 23, 26, 27], [18, 19, 20, 21], [24, 25, 26, 27], [12, 13, 14, 15], [16, 17, 18, 19]):
    print("quad", hs, measure(base[:4] + hs), flush=True)
for hs in ([20, 21, 22, 23, 24, 25, 26, 27], [12, 13, 14, 15, 16, 17, 18, 19], [11, 13, 15, 17, 19, 21, 23, 25], [10, 12, 14, 16, 18, 22, 26, 27]):
    print("all", hs, measure(hs), flush=True)
