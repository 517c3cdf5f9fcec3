#!/usr/bin/env python3
"""Step time of the bench workload family over several seeds (planning knobs from the environment, e.g.
QSIM_TILE_COMMUTE_FUSE=0): total ms, passes, ms per pass -- the plan changes which index bits meet in a tile, and with
them the memory pattern, so ONE circuit cannot judge a planner change.   python tools/step_times.py [n] [seed ...]"""
import os
import sys
import time
from pathlib import Path

# planning knobs exist in the probe build only
os.environ.setdefault("QSIM_LIBRARY", str(Path(__file__).resolve().parent.parent / "quantum_simulations_amd" / "libqsim_hip_probes.so"))
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from quantum_simulations_amd.circuits import random_1q_cx_circuit  # noqa: E402
from quantum_simulations_amd.runner.engine import make_engine  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 28
seeds = [int(s) for s in sys.argv[2:]] or [20260228, 1, 2, 3, 4, 5, 6, 7]
eng = make_engine(n)
tot_ms = tot_p = 0
for seed in seeds:
    eng.init_zero_state()
    plan = eng.plan(random_1q_cx_circuit(n, depth=40, seed=seed))
    eng.execute(plan); eng.barrier()
    reps = 3 if n >= 27 else (10 if n >= 25 else 40)
    t0 = time.perf_counter()
    for _ in range(reps):
        eng.execute(plan)
    eng.barrier()
    ms = (time.perf_counter() - t0) / reps * 1e3
    p = eng.passes_per_step(plan)
    tot_ms += ms; tot_p += p
    print(f"seed {seed}: {ms:7.3f} ms  {p} passes  {ms / p:.4f} ms/pass", flush=True)
print(f"total {tot_ms:.3f} ms  {tot_p} passes  {tot_ms / tot_p:.4f} ms/pass")
eng.close()
