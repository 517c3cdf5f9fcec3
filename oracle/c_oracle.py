"""TEST INFRASTRUCTURE ONLY -- ctypes binding of oracle/liborc.so (qsim_oracle.c), the plain-C
restatement of the reference's dense butterflies.  Used by tests and by bench.py's
`cpu_baseline` leg; never by the product package."""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

from oracle import dense_oracle

HERE = Path(__file__).resolve().parent
LIB = HERE / "liborc.so"
_lib = None


def load(build_if_missing: bool = True) -> C.CDLL:
    global _lib
    if _lib is None:
        if not LIB.exists() and build_if_missing:
            subprocess.run(["make", "-C", str(HERE)], check=True, capture_output=True)
        _lib = C.CDLL(str(LIB))
        _lib.orc_num_threads.restype = C.c_int
        for name in ("orc_apply_1q", "orc_apply_2q", "orc_apply_1q_pair", "orc_run_ops"):
            getattr(_lib, name).restype = C.c_int
        # OpenMP's default is every logical CPU of the HOST; a container with a smaller share (the GPU box: 16 cores of a
        # few hundred) then runs hundreds of spinning threads on its few cores and every gate's barrier takes milliseconds
        # (a 1 400-gate circuit at 14 qubits: minutes instead of a second).  Start from the share this process really has.
        _lib.orc_set_threads(C.c_int(max(1, min(_lib.orc_num_threads(), host_core_share()))))
    return _lib


def host_core_share() -> int:
    """Cores this process may really use: min(cpu_count, affinity, cgroup cpu.max quota)."""
    import os
    cores = os.cpu_count() or 1
    try:
        cores = min(cores, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        quota, period = Path("/sys/fs/cgroup/cpu.max").read_text().split()
        if quota != "max":
            cores = min(cores, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return min(cores, 32)


def _p(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def num_threads() -> int:
    return load().orc_num_threads()


def set_threads(n: int) -> None:
    load().orc_set_threads(C.c_int(n))


def _check(rc: int, what: str) -> None:
    if rc == -2:
        raise NotImplementedError(f"{what}: non-local qubit")
    if rc:
        raise ValueError(f"{what}: rc={rc}")


def apply_1q(psi: np.ndarray, q: int, U: np.ndarray) -> None:
    assert psi.dtype == np.complex128 and psi.flags.c_contiguous
    m = np.ascontiguousarray(U, dtype=np.complex128)
    _check(load().orc_apply_1q(_p(psi), C.c_int(len(psi).bit_length() - 1), C.c_int(q), _p(m)), "apply_1q")


def apply_2q(psi: np.ndarray, qa: int, qb: int, U: np.ndarray) -> None:
    assert psi.dtype == np.complex128 and psi.flags.c_contiguous
    m = np.ascontiguousarray(U, dtype=np.complex128)
    _check(load().orc_apply_2q(_p(psi), C.c_int(len(psi).bit_length() - 1), C.c_int(qa), C.c_int(qb),
                               _p(m)), "apply_2q")


def apply_1q_pair(c0: np.ndarray, c1: np.ndarray, U: np.ndarray) -> None:
    m = np.ascontiguousarray(U, dtype=np.complex128)
    _check(load().orc_apply_1q_pair(_p(c0), _p(c1), C.c_int(len(c0).bit_length() - 1), _p(m)), "pair")


def pack_circuit(circuit_dict: dict):
    """Circuit dict -> (nq, qubits, mats) arrays for orc_run_ops (matrices from the oracle's
    own gate table)."""
    gates = circuit_dict["gates"]
    n = len(gates)
    nq = np.zeros(n, dtype=np.int32)
    qs = np.zeros(2 * n, dtype=np.int32)
    mats = np.zeros((n, 16), dtype=np.complex128)
    for i, entry in enumerate(gates):
        name, params, qubits = dense_oracle.decode_gate(entry)
        U = dense_oracle.gate_matrix(name, params).reshape(-1)
        nq[i] = len(qubits)
        qs[2 * i: 2 * i + len(qubits)] = qubits
        mats[i, : U.size] = U
    return nq, qs, mats


def simulate(circuit_dict: dict, psi: np.ndarray | None = None) -> np.ndarray:
    """ref_dense.simulate restated in C: |0..0>, gates in list order."""
    n = circuit_dict["number_of_qubits"]
    nq, qs, mats = pack_circuit(circuit_dict)
    if psi is None:
        psi = np.empty(1 << n, dtype=np.complex128)
    _check(load().orc_run_ops(_p(psi), C.c_int(n), C.c_int(len(nq)), _p(nq), _p(qs), _p(mats),
                              C.c_int(1)), "run_ops")
    return psi
