"""Circuit-dict contract: validation, name decoding, levelisation.

Host-side mirror of the reference front end (wenbo_engine/circuit/io.py:12-117,
docs/circuit_contract.md).  Same accepted inputs, same normalised output, same
ValueError messages (pinned by the reference's test_contract.py:29-78).

Endianness is LITTLE: qubit q is bit q of the amplitude index.
"""
from __future__ import annotations

import re
from typing import Any

ENDIANNESS = "little"

# gate -> (arity, {param: python type or "array"})
_GATE_SPEC: dict[str, tuple[int, dict[str, Any]]] = {
    "H": (1, {}), "X": (1, {}), "Y": (1, {}), "Z": (1, {}), "S": (1, {}), "T": (1, {}),
    "RY": (1, {"theta": float}), "R": (1, {"k": int}), "G": (1, {"p": int}),
    "CNOT": (2, {}), "SWAP": (2, {}), "CZ": (2, {}), "CY": (2, {}),
    "CR": (2, {"k": int}), "CU": (2, {"U": "array", "exponent": int}),
}
ALL_1Q = frozenset(g for g, (a, _) in _GATE_SPEC.items() if a == 1)
ALL_2Q = frozenset(g for g, (a, _) in _GATE_SPEC.items() if a == 2)
ALL_GATES = ALL_1Q | ALL_2Q

_TOP_KEYS = {"number_of_qubits", "gates"}
_GATE_KEYS = {"qubits", "gate", "params"}
_NAME_ENCODED = re.compile(r"^(CR|R)(\d+)$")


def decode_gate_name(raw: str) -> tuple[str, dict]:
    """'CR3' -> ('CR', {'k': 3}); 'R3' -> ('R', {'k': 3}); anything else unchanged."""
    m = _NAME_ENCODED.match(raw)
    if m:
        return m.group(1), {"k": int(m.group(2))}
    return raw, {}


def _normalise_gate(entry: Any, n_qubits: int, position: int) -> dict:
    where = f"gate[{position}]"
    if not isinstance(entry, dict):
        raise ValueError(f"{where}: must be a dict")
    keys = set(entry)
    if not {"qubits", "gate"} <= keys:
        raise ValueError(f"{where}: missing 'qubits' or 'gate'")
    if keys - _GATE_KEYS:
        raise ValueError(f"{where}: unknown keys {keys - _GATE_KEYS}")

    qubits = entry["qubits"]
    if not isinstance(qubits, list) or any(not isinstance(q, int) for q in qubits):
        raise ValueError(f"{where}: qubits must be list[int]")
    for q in qubits:
        if not 0 <= q < n_qubits:
            raise ValueError(f"{where}: qubit {q} out of range [0, {n_qubits})")

    name, implied = decode_gate_name(entry["gate"])
    if name not in _GATE_SPEC:
        raise ValueError(f"{where}: unsupported gate '{entry['gate']}'")
    arity, param_spec = _GATE_SPEC[name]
    if len(qubits) != arity:
        raise ValueError(f"{where}: {name} needs {arity} qubit(s), got {len(qubits)}")

    params = dict(implied)
    params.update(entry.get("params") or {})
    for key, kind in param_spec.items():
        if key not in params:
            raise ValueError(f"{where}: {name} requires param '{key}'")
        if kind != "array" and not isinstance(params[key], (kind, int)):
            raise ValueError(f"{where}: param '{key}' bad type")
    return {"qubits": list(qubits), "gate": name, "params": params}


def validate_circuit_dict(d: dict[str, Any]) -> dict:
    """Validate and normalise a circuit dict; raises ValueError on bad input."""
    if not isinstance(d, dict):
        raise ValueError("circuit must be a dict")
    missing = _TOP_KEYS - set(d)
    if missing:
        raise ValueError(f"missing required keys: {missing}")
    extra = set(d) - _TOP_KEYS
    if extra:
        raise ValueError(f"unknown top-level keys: {extra}")
    n = d["number_of_qubits"]
    if not isinstance(n, int) or n < 1:
        raise ValueError(f"number_of_qubits must be positive int, got {n!r}")
    if not isinstance(d["gates"], list):
        raise ValueError("gates must be a list")
    return {
        "number_of_qubits": n,
        "gates": [_normalise_gate(g, n, i) for i, g in enumerate(d["gates"])],
    }


def levelize(circuit_dict: dict) -> list[list[dict]]:
    """ASAP levels: a gate lands one level after the latest gate sharing a qubit
    (reference io.py:106-117).  Gates inside a level commute trivially."""
    levels: list[list[dict]] = []
    next_free: dict[int, int] = {}
    for gate in circuit_dict["gates"]:
        lvl = max((next_free.get(q, 0) for q in gate["qubits"]), default=0)
        if lvl >= len(levels):
            levels.extend([] for _ in range(lvl + 1 - len(levels)))
        levels[lvl].append(gate)
        for q in gate["qubits"]:
            next_free[q] = lvl + 1
    return levels
