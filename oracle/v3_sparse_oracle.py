"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the v3 Spark amplitude worker.

The v3 engine keeps the state as sparse rows (idx, real, imag) and applies a gate as
  join rows x gate-matrix rows on col == bits(idx)  ->  new_idx = idx with the bits replaced by
  the matrix row  ->  amp' = g * amp  ->  SUM ... GROUP BY new_idx  ->  drop rows with
  |re| <= 1e-15 and |im| <= 1e-15
(/root/reference/v3_hisvsim_spark/src/parallel_gate_applicator.py:412-510; same arithmetic in
v2_common/gate_applicator.py:155-375).  Independent 1-qubit gates of a level are fused into one
2^k x 2^k coefficient list M[out, in] = prod_i U_i[out_i, in_i], bit i <-> i-th smallest qubit,
entries with |coef| <= 1e-15 dropped (:169-204), applied by the same join/group-by (:315-385);
2-qubit gates of a group run one by one (:97-124).  The 2-qubit sub-index is little-endian,
bit(q0) | bit(q1) << 1, with the matrix permuted to match (v2_common gates, `_to_little_endian`).
Levels = ASAP topological levels (hisvsim/partition_adapter.py:132-183); grouping =
driver._group_independent_gates (driver.py:336-367).

Spark itself cannot run here (no JVM / pyspark): parity of THIS restatement is pinned through the
reference's own v3 == v1 contract (tests/test_v3_vs_v1_direct.py:108-175, atol = rtol = 1e-10):
tests/test_oracle_v3_sparse.py checks it against the golden v1 / ref_dense states.
"""
from __future__ import annotations

import numpy as np

from oracle import dense_oracle

PRUNE = 1e-15
_LE = np.array([0, 2, 1, 3])   # big-endian pair index -> little-endian pair index


def _group_sum(new_idx: np.ndarray, amp: np.ndarray):
    order = np.argsort(new_idx, kind="stable")
    idx_sorted, amp_sorted = new_idx[order], amp[order]
    uniq, start = np.unique(idx_sorted, return_index=True)
    summed = np.add.reduceat(amp_sorted, start) if len(uniq) else amp_sorted[:0]
    keep = (np.abs(summed.real) > PRUNE) | (np.abs(summed.imag) > PRUNE)
    return uniq[keep], summed[keep]


def _apply_entries(idx, amp, qubits, entries):
    """entries: [(in_pattern, out_pattern, coef)]; pattern bit i <-> qubits[i]."""
    pattern = np.zeros_like(idx)
    cleared = idx.copy()
    for i, q in enumerate(qubits):
        pattern |= ((idx >> q) & 1) << i
        cleared &= ~(np.int64(1) << q)
    out_idx, out_amp = [], []
    for pin, pout, coef in entries:
        sel = pattern == pin
        if not sel.any():
            continue
        placed = cleared[sel].copy()
        for i, q in enumerate(qubits):
            placed |= np.int64((pout >> i) & 1) << q
        out_idx.append(placed)
        out_amp.append(coef * amp[sel])
    if not out_idx:
        return idx[:0], amp[:0]
    return _group_sum(np.concatenate(out_idx), np.concatenate(out_amp))


def _matrix_entries(M: np.ndarray):
    return [(c, r, M[r, c]) for c in range(M.shape[1]) for r in range(M.shape[0]) if abs(M[r, c]) > PRUNE]


def apply_one_qubit_gate(idx, amp, q: int, U: np.ndarray):
    return _apply_entries(idx, amp, [q], _matrix_entries(U))


def apply_two_qubit_gate(idx, amp, q0: int, q1: int, U_big_endian: np.ndarray):
    return _apply_entries(idx, amp, [q0, q1], _matrix_entries(U_big_endian[np.ix_(_LE, _LE)]))


def tensor_product_single_qubits(qubits: list[int], mats: dict) -> list:
    """Coefficient list of the fused block; `qubits` ascending."""
    k = len(qubits)
    entries = []
    for pin in range(1 << k):
        for pout in range(1 << k):
            coef = complex(1.0, 0.0)
            for i, q in enumerate(qubits):
                coef *= mats[q][(pout >> i) & 1, (pin >> i) & 1]
            if abs(coef) > PRUNE:
                entries.append((pin, pout, coef))
    return entries


def levels_of(gates: list) -> list[list[int]]:
    free: dict[int, int] = {}
    levels: list[list[int]] = []
    for gi, (_, _, qubits) in enumerate(gates):
        lvl = max((free.get(q, 0) for q in qubits), default=0)
        while len(levels) <= lvl:
            levels.append([])
        levels[lvl].append(gi)
        for q in qubits:
            free[q] = lvl + 1
    return levels


def run_circuit(circuit_dict: dict, parallel: bool = True):
    """-> (idx int64[], amp complex128[]) sparse final state, rows sorted by idx."""
    gates = [dense_oracle.decode_gate(g) for g in circuit_dict["gates"]]
    idx = np.array([0], dtype=np.int64)
    amp = np.array([1.0 + 0j], dtype=np.complex128)
    for level in levels_of(gates):
        groups, cur, used = [], [], set()
        for gi in level:                       # driver._group_independent_gates
            qs = set(gates[gi][2])
            if qs & used:
                groups.append(cur)
                cur, used = [gi], set(qs)
            else:
                cur.append(gi)
                used |= qs
        if cur:
            groups.append(cur)
        if not parallel:
            groups = [[gi] for gi in level]
        for group in groups:
            ones = [gi for gi in group if len(gates[gi][2]) == 1]
            twos = [gi for gi in group if len(gates[gi][2]) == 2]
            if len(ones) > 1:
                mats = {gates[gi][2][0]: dense_oracle.gate_matrix(gates[gi][0], gates[gi][1]) for gi in ones}
                qs = sorted(mats)
                idx, amp = _apply_entries(idx, amp, qs, tensor_product_single_qubits(qs, mats))
            elif ones:
                name, params, (q,) = gates[ones[0]]
                idx, amp = apply_one_qubit_gate(idx, amp, q, dense_oracle.gate_matrix(name, params))
            for gi in twos:
                name, params, (q0, q1) = gates[gi]
                idx, amp = apply_two_qubit_gate(idx, amp, q0, q1, dense_oracle.gate_matrix(name, params))
    return idx, amp


def to_dense(idx, amp, n: int) -> np.ndarray:
    psi = np.zeros(1 << n, dtype=np.complex128)
    psi[idx] = amp
    return psi
