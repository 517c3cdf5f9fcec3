#!/usr/bin/env python3
"""One process = one memory-pattern configuration of the fused tile pass, for `rocprofv3 --pmc`:
gate-less passes (load -> LDS -> store; QSIM_DEBUG_SKIP_GATES=4) over the given 8 high tile bits, or
the device-to-device copy kernel as the reference pattern.  Probe build only.

    python3 tools/tile_pass_once.py <n_qubits> <b0,b1,...,b7 | copy | bench> [reps]

`bench` runs the default bench circuit's passes WITH their gates (the product plan).  Prints the
median milliseconds per launch (HIP events).  The knobs are read by the library at load time, so they
are put into the environment here, before the import (rocprofv3 needs the program itself after `--`:
no env / bash -c wrapper)."""
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
_PROBES = os.path.join(_HERE, "..", "quantum_simulations_amd", "libqsim_hip_probes.so")
os.environ.setdefault("QSIM_LIBRARY", os.path.abspath(_PROBES))
n = int(sys.argv[1])
what = sys.argv[2]
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
if what not in ("copy", "bench"):
    os.environ["QSIM_DEBUG_SKIP_GATES"] = "4"
    os.environ["QSIM_DEBUG_TILE_BITS"] = what
sys.path.insert(0, os.path.join(_HERE, ".."))
import numpy as np  # noqa: E402

from quantum_simulations_amd.kernel import gates as gt  # noqa: E402
from quantum_simulations_amd.kernel.device import DeviceChunk  # noqa: E402

dev = DeviceChunk.empty(n)
dev.init_random(1)
ts = []
if what == "copy":
    other = DeviceChunk.empty(n)
    other.copy_from(dev)
    dev.sync()
    other.sync()
    for _ in range(reps):
        other.time_begin()
        other.copy_from(dev)
        ts.append(other.time_end())
elif what == "bench":
    from quantum_simulations_amd.circuits import random_1q_cx_circuit
    from quantum_simulations_amd.runner.engine import SingleGpuEngine
    eng = SingleGpuEngine.__new__(SingleGpuEngine)
    eng.n, eng.mode, eng.state = n, "fused", dev
    plan = eng.plan(random_1q_cx_circuit(n, depth=40), repeats=reps)
    eng.execute(plan)
    dev.sync()
    for _ in range(max(1, reps // 3)):
        dev.time_begin()
        eng.execute(plan)
        ts.append(dev.time_end() / eng.last_passes)
else:
    ops = [([3], gt.H()), ([4], gt.H())]
    dev.apply_ops(ops)
    dev.sync()
    for _ in range(reps):
        dev.time_begin()
        dev.apply_ops(ops)
        ts.append(dev.time_end())
print("ms_per_launch %.4f" % float(np.median(ts)), flush=True)
