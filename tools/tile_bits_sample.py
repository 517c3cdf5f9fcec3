#!/usr/bin/env python3
"""Gate-less fused pass (load -> store of every tile, probe build) timed for MANY sets of the 8 high tile bits in one
process: random sets plus structured ones; one line per set "b0 ... b7 ms".  Raw material for a memory-pattern model of
the pass builder.   python tools/tile_bits_sample.py [n_qubits] [n_random_sets] [seed] > gpurun_out/tile_bits_samples.txt"""
import os
import sys
import time

_PROBES = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "quantum_simulations_amd", "libqsim_hip_probes.so")
os.environ.setdefault("QSIM_LIBRARY", os.path.abspath(_PROBES))
os.environ["QSIM_DEBUG_SKIP_GATES"] = "4"
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np  # noqa: E402

from quantum_simulations_amd.kernel import gates as gt  # noqa: E402
from quantum_simulations_amd.kernel.device import DeviceChunk  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 28
count = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
dev = DeviceChunk.empty(n)
dev.init_random(1)
ops = [([3], gt.H()), ([4], gt.H())]
rng = np.random.default_rng(int(sys.argv[3]) if len(sys.argv) > 3 else 11)
sets = [sorted(int(b) for b in rng.choice(np.arange(3, n), size=8, replace=False)) for _ in range(count)]
t0 = time.time()
for i, bits in enumerate(sets):
    os.environ["QSIM_DEBUG_TILE_BITS"] = ",".join(map(str, bits))
    dev.apply_ops(ops)
    dev.sync()
    ts = []
    for _ in range(3):
        dev.time_begin()
        dev.apply_ops(ops)
        ts.append(dev.time_end())
    print(*bits, f"{min(ts):.4f}", flush=True)
    if i % 200 == 199:
        print(f"# {i + 1} sets, {time.time() - t0:.0f} s", file=sys.stderr, flush=True)
