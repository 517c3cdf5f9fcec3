"""Qiskit front-end: a (transpiled) `QuantumCircuit` -> circuit dict.

Interface of wenbo_engine/circuit/import_qiskit.py:3-36: `SUPPORTED_BASIS` is the basis to
transpile to, `qiskit_to_dict(qc)` walks `qc.data` and needs nothing from qiskit itself -- any
object with `num_qubits`, `data` (items with `.operation.name`, `.operation.params`, `.qubits`)
and `find_bit(q).index` works, which is how the tests drive it without qiskit installed.
Barriers, measurements, resets, delays and identities are dropped; anything else outside the
basis raises ValueError naming the gate.
"""
from __future__ import annotations

SUPPORTED_BASIS = ["h", "x", "y", "z", "s", "t", "ry", "cx", "cz", "swap", "cy"]

_GATE_OF = {"h": "H", "x": "X", "y": "Y", "z": "Z", "s": "S", "t": "T", "ry": "RY",
            "cx": "CNOT", "cnot": "CNOT", "swap": "SWAP", "cz": "CZ", "cy": "CY"}
_IGNORED = ("barrier", "measure", "reset", "delay", "id")


def qiskit_to_dict(qc) -> dict:
    gates = []
    for item in qc.data:
        op = item.operation
        name = op.name.lower()
        if name in _IGNORED:
            continue
        if name not in _GATE_OF:
            raise ValueError(f"Unsupported gate '{name}'. Transpile to basis {SUPPORTED_BASIS} first.")
        params = {"theta": float(op.params[0])} if name == "ry" else {}
        gates.append({"qubits": [qc.find_bit(q).index for q in item.qubits],
                      "gate": _GATE_OF[name], "params": params})
    return {"number_of_qubits": qc.num_qubits, "gates": gates}
