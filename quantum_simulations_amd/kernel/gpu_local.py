"""HIP kernel module with the reference's local-kernel interface.

Drop-in for `wenbo_engine.kernel.cpu_scalar` / `cpu_batched` (cpu_scalar.py:13-47,
cpu_batched.py:12-40): `apply_1q(chunk, qubit, U)`, `apply_2q(chunk, qa, qb, U)` and
`check_local(qubit, chunk_len)` with the same argument meaning, in-place semantics,
`None` return and `NotImplementedError("... non-local ...")` behaviour.

`chunk` may be
  * a `DeviceChunk` (HBM-resident; the fast path -- nothing crosses PCIe), or
  * a caller-owned 1-D numpy array (complex64 or complex128, power-of-two length):
    it is uploaded, updated on the GPU in complex128, and written back in place in its
    own dtype -- exactly what the reference's numpy expressions produce.
All arithmetic runs in libqsim_hip.so; there is no CPU path.
"""
from __future__ import annotations

import math

import numpy as np

from quantum_simulations_amd.kernel.device import DeviceChunk


def check_local(qubit: int, chunk_len: int) -> None:
    k = int(math.log2(chunk_len))
    if qubit >= k:
        raise NotImplementedError(
            f"qubit {qubit} >= log2(chunk_size)={k}: non-local gate requires layout/collect step")


def _host_round_trip(chunk: np.ndarray, fn) -> None:
    if chunk.ndim != 1:
        raise ValueError("chunk must be 1-D")
    dev = DeviceChunk.from_numpy(chunk)
    try:
        fn(dev)
        chunk[:] = dev.download()
    finally:
        dev.close()


def apply_1q(chunk, qubit: int, U: np.ndarray) -> None:
    check_local(qubit, len(chunk))
    if isinstance(chunk, DeviceChunk):
        chunk.apply_1q(qubit, U)
    else:
        _host_round_trip(chunk, lambda d: d.apply_1q(qubit, U))


def apply_2q(chunk, qa: int, qb: int, U: np.ndarray) -> None:
    check_local(qa, len(chunk))
    check_local(qb, len(chunk))
    if isinstance(chunk, DeviceChunk):
        chunk.apply_2q(qa, qb, U)
    else:
        _host_round_trip(chunk, lambda d: d.apply_2q(qa, qb, U))


def apply_ops(chunk, ops) -> None:
    """A whole pass `[(qubits, U), ...]` in order (single_node._process_local_chunk,
    single_node.py:208-216) with one upload/download for numpy chunks."""
    for qubits, _ in ops:
        for q in qubits:
            check_local(q, len(chunk))
    if isinstance(chunk, DeviceChunk):
        chunk.apply_ops(ops)
    else:
        _host_round_trip(chunk, lambda d: d.apply_ops(ops))
