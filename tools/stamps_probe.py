#!/usr/bin/env python3
"""In-kernel cycle stamps of k_tile (probe build, QSIM_DEBUG_STAMPS=<file>): per sampled workgroup the
time from entry to "tile in LDS", the gate engine, and the store issue; plus how many workgroups were alive
at once.   python tools/stamps_probe.py [n]      (run with the probe library copied over libqsim_hip.so)"""
import os
import sys

# probe build of the library (make -C quantum_simulations_amd/csrc probes): the product build reads no probe knobs
_PROBES = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "quantum_simulations_amd", "libqsim_hip_probes.so")
os.environ.setdefault("QSIM_LIBRARY", os.path.abspath(_PROBES))
from pathlib import Path

import numpy as np

out = "/tmp/qsim_stamps.txt"
if os.path.exists(out):
    os.remove(out)
os.environ["QSIM_DEBUG_STAMPS"] = out
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from quantum_simulations_amd.circuits import random_1q_cx_circuit  # noqa: E402
from quantum_simulations_amd.runner.engine import make_engine  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 28
eng = make_engine(n)
eng.init_zero_state()
plan = eng.plan(random_1q_cx_circuit(n, depth=40))
eng.execute(plan)
eng.barrier()
eng.close()
passes, cur = [], None
for line in open(out):
    if line.startswith("#"):
        cur = []
        passes.append((line.strip(), cur))
    else:
        cur.append([int(x) for x in line.split()])
for title, rows in passes:
    r = np.array(rows, dtype=np.float64)
    if not len(r):
        continue
    load, eng_t, store = r[:, 1] - r[:, 0], r[:, 2] - r[:, 1], r[:, 3] - r[:, 2]
    span = r[:, 3].max() - r[:, 0].min()
    print(f"{title}: sampled {len(r)}  load {np.median(load):7.0f}  engine {np.median(eng_t):7.0f}  store-issue {np.median(store):6.0f}  "
          f"lifetime {np.median(r[:, 3] - r[:, 0]):7.0f}  kernel span {span:9.0f} ticks")
