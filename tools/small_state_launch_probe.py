#!/usr/bin/env python3
"""Is a small-state step launch bound (would a hipGraph of the cached plan pay)?  Wall time per step of the bench circuit at
12-24 qubits (200 repeated executions, plans cached) against the sum of its kernels' own times (HIP events).
    python tools/small_state_launch_probe.py"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from quantum_simulations_amd import circuits as gen
from quantum_simulations_amd.runner.engine import SingleGpuEngine
for n in (12, 16, 18, 20, 22, 24):
    e = SingleGpuEngine(n, device=0, mode="fused", layout="identity")
    cd = gen.random_1q_cx_circuit(n, depth=40)
    e.init_zero_state()
    reps = 200
    plan = e.plan(cd, repeats=reps + 20)
    for _ in range(20): e.execute(plan)
    e.barrier()
    t = time.perf_counter()
    for _ in range(reps): e.execute(plan)
    e.barrier()
    wall = (time.perf_counter() - t) / reps * 1e3
    e2 = SingleGpuEngine(n, device=0, mode="fused", layout="identity"); e2.init_zero_state(); p2 = e2.plan(cd, repeats=12)
    for _ in range(2): e2.execute(p2)
    e2.barrier(); e2.profile_begin()
    for _ in range(10): e2.execute(p2)
    e2.barrier(); prof = e2.profile_end()
    kern = sum(x["total_ms"] for x in prof) / 10
    launches = sum(x["launches"] for x in prof) / 10
    print(f"n={n}: {wall:.4f} ms/step wall, {kern:.4f} ms of kernels in {launches:.0f} launches ({wall/launches*1e3:.1f} us per launch wall, {kern/launches*1e3:.1f} us kernel)", flush=True)
    e.close(); e2.close()
