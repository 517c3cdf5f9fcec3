#!/usr/bin/env python3
"""Offline (no GPU): re-layouts and HBM passes of rank 0's schedule under every staging method -- the Atlas heuristic, its ILP
(scipy.optimize.milp, 30 s per solve like the reference; a solve that runs out of time counts as infeasible and the
heuristic's sets are used), farthest-next-use stage sets ("belady") and stage boundaries planned together with the tile
passes ("tiles", runner/partition_plan.py) -- for BASELINE configs 4 / 5 at full size.
    python tools/staging_methods_table.py N_QUBITS N_RANKS [methods...]"""
import sys
import time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from quantum_simulations_amd.circuit.io import validate_circuit_dict
from quantum_simulations_amd.circuits import generate_ghz_circuit, generate_ghz_qft, random_1q_cx_circuit, random_clifford_t_circuit
from quantum_simulations_amd.runner.distributed import DistributedEngine, PlanningBackend

n, world = int(sys.argv[1]), int(sys.argv[2])
methods = sys.argv[3:] or ["heuristic", "belady", "tiles", "ilp"]
p = world.bit_length() - 1
for name, cd in (("config 4: Clifford+T depth 60", random_clifford_t_circuit(n, depth=60)), ("config 5: GHZ", generate_ghz_circuit(n)),
                 ("config 5: GHZ+QFT", generate_ghz_qft(n)), ("bench: random 1q+CX depth 40", random_1q_cx_circuit(n, depth=40))):
    cd = validate_circuit_dict(cd)
    for method in methods:
        eng = DistributedEngine(n, world, 0, backend=PlanningBackend(n - p), init_process_group=False, layout="identity", staging_method=method)
        eng.init_zero_state()
        t0 = time.perf_counter()
        plan = eng.plan(cd)
        dt = time.perf_counter() - t0
        eng.relayout_log = []
        eng.execute(plan)
        print(f"{n} qubits / {world} ranks | {name} ({len(cd['gates'])} gates) | {method:9s} | re-layouts {len(eng.relayout_log)} (m = {eng.relayout_log}) | "
              f"HBM passes {eng.last_passes} | planning {dt:.1f} s", flush=True)
