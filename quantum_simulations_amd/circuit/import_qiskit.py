"""Qiskit front-end: a (transpiled) `QuantumCircuit` -> circuit dict.

Interface of wenbo_engine/circuit/import_qiskit.py:3-36: `SUPPORTED_BASIS` is the basis to
transpile to, `qiskit_to_dict(qc)` walks `qc.data` and needs nothing from qiskit itself -- any
object with `num_qubits`, `data` (items with `.operation.name`, `.operation.params`, `.qubits`)
and `find_bit(q).index` works, which is how the tests drive it without qiskit installed.
Barriers, measurements, resets, delays and identities are dropped; anything else outside the
basis raises ValueError naming the gate.
"""
from __future__ import annotations

# (qiskit instruction name, gate of the circuit contract, name of its angle parameter)
_TABLE = (("h", "H", None), ("x", "X", None), ("y", "Y", None), ("z", "Z", None), ("s", "S", None),
          ("t", "T", None), ("ry", "RY", "theta"), ("cx", "CNOT", None), ("cz", "CZ", None),
          ("swap", "SWAP", None), ("cy", "CY", None))
SUPPORTED_BASIS = [name for name, _, _ in _TABLE]
_GATE_OF = {name: (gate, angle) for name, gate, angle in _TABLE}
_GATE_OF["cnot"] = _GATE_OF["cx"]
_NOT_GATES = ("barrier", "measure", "reset", "delay", "id")


def qiskit_to_dict(qc) -> dict:
    gates = []
    for item in qc.data:
        name = item.operation.name.lower()
        if name in _NOT_GATES:
            continue
        if name not in _GATE_OF:
            raise ValueError(f"Unsupported gate '{name}'. Transpile to basis {SUPPORTED_BASIS} first.")
        gate, angle = _GATE_OF[name]
        gates.append({"qubits": [qc.find_bit(q).index for q in item.qubits], "gate": gate,
                      "params": {angle: float(item.operation.params[0])} if angle else {}})
    return {"number_of_qubits": qc.num_qubits, "gates": gates}
