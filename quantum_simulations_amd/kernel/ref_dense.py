"""Gate-by-gate dense simulation, interface of `wenbo_engine.kernel.ref_dense.simulate`
(ref_dense.py:44-57): validate, start from |0..0>, apply gates in list order with no
levelling and no fusion, return the complex128 vector.  Here every gate is one HIP
launch on an HBM-resident state (the CPU restatement used as the test oracle lives in
`oracle/`, not in this package).
"""
from __future__ import annotations

import numpy as np

from quantum_simulations_amd.circuit.io import validate_circuit_dict
from quantum_simulations_amd.kernel import gates as gate_table
from quantum_simulations_amd.kernel.device import DeviceChunk


def simulate_on_device(circuit_dict: dict, device: int = 0) -> DeviceChunk:
    cd = validate_circuit_dict(circuit_dict)
    psi = DeviceChunk.zero_state(cd["number_of_qubits"], device)
    for g in cd["gates"]:
        U = gate_table.gate_matrix(g["gate"], g["params"])
        if len(g["qubits"]) == 1:
            psi.apply_1q(g["qubits"][0], U)
        else:
            psi.apply_2q(g["qubits"][0], g["qubits"][1], U)
    return psi


def simulate(circuit_dict: dict, device: int = 0) -> np.ndarray:
    psi = simulate_on_device(circuit_dict, device)
    try:
        return psi.download()
    finally:
        psi.close()
