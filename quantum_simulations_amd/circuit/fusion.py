"""1-qubit gate fusion and level batching.

Host-side planner with the behaviour of the reference's
wenbo_engine/circuit/fusion.py:41-165.  On the MI355X every *pass* emitted here is
one HBM round trip of the shard (one fused-tile launch sequence) instead of one
disk read/write cycle, so the same two ideas pay the same way:

  * `fuse_1q_ops`  -- runs of 1-qubit gates on one qubit collapse to a single 2x2
    (composition is `U_new @ U_old`), a 2-qubit gate flushes its qubits first;
  * `batch_levels` -- consecutive levels whose gates are all shard-local are merged
    into one pass; a level with a non-local gate is always its own pass.

Ops are `(qubits, U)` tuples with `qubits` a list of ints and `U` a complex128
ndarray (2x2 or 4x4, pair big-endian).
"""
from __future__ import annotations

import numpy as np

from quantum_simulations_amd.kernel import gates as gate_table

Op = tuple  # (list[int], np.ndarray)


def split_by_locality(level_gates: list[dict], k: int) -> tuple[list[Op], list[Op]]:
    """Gate dicts of one level -> (local ops, non-local ops) for 2^k-amplitude shards."""
    inside: list[Op] = []
    outside: list[Op] = []
    for gate in level_gates:
        op = (gate["qubits"], gate_table.gate_matrix(gate["gate"], gate["params"]))
        (inside if max(gate["qubits"]) < k else outside).append(op)
    return inside, outside


def fuse_1q_ops(ops: list[Op]) -> list[Op]:
    """Collapse consecutive 1-qubit gates per qubit; order of emission matches the
    reference (2q gate flushes its own qubits in its qubit order; leftovers are
    emitted by ascending qubit at the end)."""
    if not ops:
        return ops
    waiting: dict[int, np.ndarray] = {}
    fused: list[Op] = []
    for qubits, U in ops:
        if len(qubits) == 1:
            q = qubits[0]
            waiting[q] = U @ waiting[q] if q in waiting else U.copy()
            continue
        for q in qubits:
            if q in waiting:
                fused.append(([q], waiting.pop(q)))
        fused.append((qubits, U))
    fused.extend(([q], waiting[q]) for q in sorted(waiting))
    return fused


def batch_levels(levels: list[list[dict]], k: int) -> list[dict]:
    """Levels -> passes `{"local_ops", "nonlocal_ops", "level_indices"}`."""
    passes: list[dict] = []
    run_ops: list[Op] = []
    run_levels: list[int] = []

    def close_run() -> None:
        if run_ops:
            passes.append({"local_ops": fuse_1q_ops(list(run_ops)),
                           "nonlocal_ops": [],
                           "level_indices": list(run_levels)})
            run_ops.clear()
            run_levels.clear()

    for index, gates in enumerate(levels):
        if not gates:
            continue
        inside, outside = split_by_locality(gates, k)
        if outside:
            close_run()
            passes.append({"local_ops": inside, "nonlocal_ops": outside,
                           "level_indices": [index]})
        else:
            run_ops.extend(inside)
            run_levels.append(index)
    close_run()
    return passes


def fusion_stats(levels: list[list[dict]], k: int) -> dict:
    """Before/after counts, same keys as the reference (fusion.py:145-165)."""
    passes = batch_levels(levels, k)
    n_levels = sum(1 for lv in levels if lv)
    n_passes = len(passes)
    saved = (1 - n_passes / max(n_levels, 1)) * 100
    return {
        "original_levels": n_levels,
        "fused_passes": n_passes,
        "local_only_passes": sum(1 for p in passes if not p["nonlocal_ops"]),
        "io_reduction": f"{n_levels}→{n_passes} ({saved:.0f}% fewer)",
        "ops_before": sum(len(lv) for lv in levels),
        "ops_after": sum(len(p["local_ops"]) + len(p["nonlocal_ops"]) for p in passes),
    }
