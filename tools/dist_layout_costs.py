#!/usr/bin/env python3
"""Offline (no GPU): the partition's initial-layout search (DistributedEngine.choose_initial_layout) on the four workload
families: model cost, passes and re-layouts of the staged schedule under the identity and under the chosen assignment.
    python tools/dist_layout_costs.py N_QUBITS N_RANKS"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from quantum_simulations_amd.circuits import random_1q_cx_circuit, random_clifford_t_circuit, generate_ghz_qft, generate_ghz_circuit
from quantum_simulations_amd.circuit.io import validate_circuit_dict
from quantum_simulations_amd.runner.distributed import DistributedEngine, DryBackend
n, world = int(sys.argv[1]), int(sys.argv[2])
p = world.bit_length() - 1
eng = DistributedEngine(n, world, 0, backend=DryBackend(n - p), init_process_group=False, layout="search")
for name, cd in (("rand", random_1q_cx_circuit(n, depth=40)), ("clifft", random_clifford_t_circuit(n, depth=60)), ("ghz_qft", generate_ghz_qft(n)), ("ghz", generate_ghz_circuit(n))):
    eng.init_zero_state()
    eng.plan(validate_circuit_dict(cd))
    print(n, world, name, eng.layout_info)
