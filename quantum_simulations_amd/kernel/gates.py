"""Host-side gate table for the MI355X engine.

Same gate set, names, parameters and matrix conventions as the reference
(wenbo_engine/kernel/gates.py:24-112; v1_implementation/src/gates.py:27-188):

  * 1-qubit gates are 2x2 complex128;
  * 2-qubit gates are 4x4 complex128, **big-endian inside the pair**:
    row/col = 2*bit(qubits[0]) + bit(qubits[1]); qubits[0] is the control.

The matrices are built on the host (32 B / 256 B each) and handed to the HIP
kernels as kernel arguments; the library inspects their zero/one pattern to pick
the specialised kernel (diagonal, controlled, swap) that moves the fewest HBM bytes.
"""
from __future__ import annotations


import numpy as np

_SQRT_HALF = 1.0 / np.sqrt(2.0)


def _arr(rows) -> np.ndarray:
    return np.array(rows, dtype=np.complex128)


def _lift_controlled(block: np.ndarray) -> np.ndarray:
    out = np.eye(4, dtype=np.complex128)
    out[2:4, 2:4] = block
    return out


# ---- 1-qubit -------------------------------------------------------------------
def H() -> np.ndarray:
    return _arr([[_SQRT_HALF, _SQRT_HALF], [_SQRT_HALF, -_SQRT_HALF]])


def X() -> np.ndarray:
    return _arr([[0, 1], [1, 0]])


def Y() -> np.ndarray:
    return _arr([[0, -1j], [1j, 0]])


def Z() -> np.ndarray:
    return _arr([[1, 0], [0, -1]])


def S() -> np.ndarray:
    return _arr([[1, 0], [0, 1j]])


def T() -> np.ndarray:
    return _arr([[1, 0], [0, np.exp(1j * np.pi / 4)]])


def RY(theta: float) -> np.ndarray:
    c, s = np.cos(theta / 2), np.sin(theta / 2)
    return _arr([[c, -s], [s, c]])


def R(k: int) -> np.ndarray:
    return _arr([[1, 0], [0, np.exp(2j * np.pi / 2 ** k)]])


def G(p: int) -> np.ndarray:
    keep, move = np.sqrt(1.0 / p), np.sqrt(1.0 - 1.0 / p)
    return _arr([[keep, -move], [move, keep]])


# ---- 2-qubit -------------------------------------------------------------------
def CNOT() -> np.ndarray:
    return _lift_controlled(X())


def CY() -> np.ndarray:
    return _lift_controlled(Y())


def CZ() -> np.ndarray:
    return _lift_controlled(Z())


def CR(k: int) -> np.ndarray:
    return _lift_controlled(R(k))


def CU(U, exponent: int) -> np.ndarray:
    return _lift_controlled(
        np.linalg.matrix_power(np.asarray(U, dtype=np.complex128), exponent))


def SWAP() -> np.ndarray:
    m = np.zeros((4, 4), dtype=np.complex128)
    m[0, 0] = m[1, 2] = m[2, 1] = m[3, 3] = 1
    return m


_BUILDERS = {
    # name: (arity, builder taking the params dict)
    "H": (1, lambda p: H()), "X": (1, lambda p: X()), "Y": (1, lambda p: Y()),
    "Z": (1, lambda p: Z()), "S": (1, lambda p: S()), "T": (1, lambda p: T()),
    "RY": (1, lambda p: RY(p["theta"])), "R": (1, lambda p: R(p["k"])),
    "G": (1, lambda p: G(p["p"])),
    "CNOT": (2, lambda p: CNOT()), "SWAP": (2, lambda p: SWAP()),
    "CZ": (2, lambda p: CZ()), "CY": (2, lambda p: CY()),
    "CR": (2, lambda p: CR(p["k"])),
    "CU": (2, lambda p: CU(p["U"], p["exponent"])),
}


def gate_matrix(name: str, params: dict) -> np.ndarray:
    """Unitary for a normalised gate entry (reference gates.py:92-108)."""
    try:
        _, build = _BUILDERS[name]
    except KeyError:
        raise ValueError(f"unknown gate {name}") from None
    return build(params or {})


def is_2q(name: str) -> bool:
    """Arity test (reference gates.py:111-112)."""
    entry = _BUILDERS.get(name)
    return entry is not None and entry[0] == 2
