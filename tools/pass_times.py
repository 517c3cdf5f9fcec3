#!/usr/bin/env python3
"""Per-pass timing of the bench workload (QSIM_DEBUG_STATS=2: every fused pass timed synchronously and
printed with its record count and tile bits).   python tools/pass_times.py [n_qubits] [depth] [seed ...]"""
import os
import sys
from pathlib import Path

os.environ["QSIM_DEBUG_STATS"] = "2"
# (knobs exist in the probe build only)
os.environ.setdefault("QSIM_LIBRARY", str(Path(__file__).resolve().parent.parent / "quantum_simulations_amd" / "libqsim_hip_probes.so"))
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from quantum_simulations_amd.circuits import random_1q_cx_circuit  # noqa: E402
from quantum_simulations_amd.runner.engine import make_engine  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 28
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 40
seeds = [int(s) for s in sys.argv[3:]] or [20260228]
eng = make_engine(n)
for seed in seeds:
    eng.init_zero_state()
    plan = eng.plan(random_1q_cx_circuit(n, depth=depth, seed=seed))
    eng.execute(plan)          # warm-up (prints too)
    eng.barrier()
    print(f"---- second execution seed {seed} ----", file=sys.stderr, flush=True)
    eng.execute(plan)
    eng.barrier()
eng.close()
