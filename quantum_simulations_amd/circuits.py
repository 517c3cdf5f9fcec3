"""Circuit generators (circuit-dict producers).

The named families reproduce, gate for gate, the dicts emitted by the reference
generators v1_implementation/src/circuits.py:11-88 (same in v2/v3 `v2_common`):
GHZ, QFT (no final swaps, CR on [k, j]), QPE, W, Hadamard wall, W+QFT, GHZ+QFT,
"GHZ proned".  `random_1q_cx_circuit` and `random_clifford_t_circuit` are this
build's own seeded workloads for BASELINE configs 2 and 4.
"""
from __future__ import annotations

import numpy as np

from quantum_simulations_amd.kernel import gates as gate_table


def _circuit(n: int, gates: list[dict]) -> dict:
    return {"number_of_qubits": n, "gates": gates}


def generate_ghz_circuit(num_qubits: int, reverse: bool = False) -> dict:
    gates = [{"qubits": [0], "gate": "H"}]
    gates += [{"qubits": [q - 1, q], "gate": "CNOT"} for q in range(1, num_qubits)]
    return _circuit(num_qubits, gates[::-1] if reverse else gates)


def generate_qft_circuit(num_qubits: int, reverse: bool = False) -> dict:
    gates: list[dict] = []
    for target in range(num_qubits):
        gates.append({"qubits": [target], "gate": "H"})
        for ctrl in range(target + 1, num_qubits):
            gates.append({"qubits": [ctrl, target], "gate": "CR",
                          "params": {"k": ctrl - target + 1}})
    return _circuit(num_qubits, gates[::-1] if reverse else gates)


def generate_qpe_circuit(num_qubits: int) -> dict:
    """Phase estimation of U = Z on `num_qubits` counting qubits + 1 eigenstate qubit."""
    u = np.array([[1, 0], [0, -1]], dtype=complex)
    gates = [{"qubits": [j], "gate": "H"} for j in range(num_qubits)]
    gates += [{"qubits": [j, num_qubits], "gate": "CU",
               "params": {"U": u, "exponent": 2 ** j}} for j in range(num_qubits)]
    for j in range(num_qubits):
        gates += [{"qubits": [m, j], "gate": "CR", "params": {"k": j - m + 1}}
                  for m in range(j)]
        gates.append({"qubits": [j], "gate": "H"})
    return _circuit(num_qubits + 1, gates)


def generate_w_circuit(n_qubits: int, reverse: bool = False) -> dict:
    gates = [
        {"qubits": [0], "gate": "X"},
        {"qubits": [1], "gate": "G", "params": {"p": n_qubits}},
        {"qubits": [1, 0], "gate": "CNOT"},
    ]
    for i in range(n_qubits - 2):
        p = n_qubits - 1 - i
        gates.append({"qubits": [i + 1, i + 2], "gate": "CU",
                      "params": {"U": gate_table.G(p), "exponent": 1, "name": f"CG{p}"}})
        gates.append({"qubits": [i + 2, i + 1], "gate": "CNOT"})
    return _circuit(n_qubits, gates[::-1] if reverse else gates)


def generate_hadamard_wall(n_qubits: int) -> dict:
    return _circuit(n_qubits, [{"qubits": [q], "gate": "H"} for q in range(n_qubits)])


def generate_w_qft(n_qubits: int) -> dict:
    return _circuit(n_qubits, generate_w_circuit(n_qubits)["gates"]
                    + generate_qft_circuit(n_qubits)["gates"])


def generate_ghz_qft(n_qubits: int) -> dict:
    return _circuit(n_qubits, generate_ghz_circuit(n_qubits)["gates"]
                    + generate_qft_circuit(n_qubits)["gates"])


def generate_ghz_proned(n_qubits: int, depth: int) -> dict:
    """GHZ ladders forwards/backwards alternately, truncated to `depth` gates."""
    gates: list[dict] = []
    backwards = False
    while len(gates) < depth:
        gates += generate_ghz_circuit(n_qubits, backwards)["gates"]
        backwards = not backwards
    return _circuit(n_qubits, gates[:depth])


# --------------------------------------------------------------- seeded workloads
_RANDOM_1Q = ("H", "X", "Y", "Z", "S", "T", "RY")


def random_1q_cx_circuit(n_qubits: int, depth: int = 40, seed: int = 20260228) -> dict:
    """BASELINE config 2: `depth` layers; even layers put one random 1q gate from
    {H,X,Y,Z,S,T,RY(theta~U[0,2pi))} on every qubit, odd layers put CNOTs on a random
    perfect matching (random orientation; one qubit idles when n is odd)."""
    rng = np.random.default_rng(seed)
    gates: list[dict] = []
    for layer in range(depth):
        if layer % 2 == 0:
            for q in range(n_qubits):
                name = _RANDOM_1Q[int(rng.integers(len(_RANDOM_1Q)))]
                entry: dict = {"qubits": [q], "gate": name}
                if name == "RY":
                    entry["params"] = {"theta": float(rng.uniform(0.0, 2.0 * np.pi))}
                gates.append(entry)
        else:
            order = [int(q) for q in rng.permutation(n_qubits)]
            for a, b in zip(order[0::2], order[1::2]):
                gates.append({"qubits": [a, b], "gate": "CNOT"})
    return _circuit(n_qubits, gates)


def random_clifford_t_circuit(n_qubits: int, depth: int = 60, seed: int = 20260432) -> dict:
    """BASELINE config 4: `depth` layers of n_qubits//2 gates drawn uniformly from
    {H, S, T, CNOT} on uniformly random (distinct) qubits."""
    rng = np.random.default_rng(seed)
    gates: list[dict] = []
    per_layer = max(1, n_qubits // 2)
    for _ in range(depth):
        for _ in range(per_layer):
            name = ("H", "S", "T", "CNOT")[int(rng.integers(4))]
            if name == "CNOT":
                a, b = (int(q) for q in rng.choice(n_qubits, size=2, replace=False))
                gates.append({"qubits": [a, b], "gate": "CNOT"})
            else:
                gates.append({"qubits": [int(rng.integers(n_qubits))], "gate": name})
    return _circuit(n_qubits, gates)
