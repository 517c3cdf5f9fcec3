"""TEST INFRASTRUCTURE ONLY -- stdlib-sqlite3 restatement of the v1 engine (BASELINE config 1).

Follows, from scratch, the algorithm of /root/reference/v1_implementation:
  state table (version, idx, real, imag) ........ sql/schema.sql:1-8
  gate_matrix table (name, arity, row, col, re, im) sql/schema.sql:11-19, gate_loader.py:15-56
  |0..0> = one row idx 0 ........................ src/state_manager.py:13-17
  one INSERT..SELECT..JOIN..GROUP BY per gate ... src/gate_translator.py:9-55
  2q sub-index is little-endian (bit(q0) | bit(q1)<<1) with the matrix permuted to match
  (src/gates.py:9-14); zero-amplitude rows are kept, old versions stay in the table.
Leaves out the WAL / checkpoint tables (crash recovery, not arithmetic).

Parity status: PINNED -- tests/test_oracle_v1_sql.py compares final states and per-version
row counts with tests/golden/v1_sql.npz (produced by running the reference v1 engine).
"""
from __future__ import annotations

import sqlite3

import numpy as np

from oracle import dense_oracle

_LE = np.array([0, 2, 1, 3])  # big-endian pair index -> little-endian pair index


def _connect() -> sqlite3.Connection:
    con = sqlite3.connect(":memory:")
    con.executescript("""
        CREATE TABLE state (version INT, idx BIGINT, real DOUBLE, imag DOUBLE,
                            PRIMARY KEY (version, idx));
        CREATE TABLE gate_matrix (gate_name TEXT, arity INT, row INT, col INT,
                                  real DOUBLE, imag DOUBLE,
                                  PRIMARY KEY (gate_name, arity, row, col));
    """)
    return con


def _register(con, key: str, U: np.ndarray) -> None:
    arity = 1 if U.shape == (2, 2) else 2
    M = U if arity == 1 else U[np.ix_(_LE, _LE)]
    con.executemany(
        "INSERT OR REPLACE INTO gate_matrix VALUES (?, ?, ?, ?, ?, ?)",
        [(key, arity, r, c, float(M[r, c].real), float(M[r, c].imag))
         for r in range(M.shape[0]) for c in range(M.shape[1])])


def _gate_sql(key: str, qubits: list[int], v: int) -> str:
    if len(qubits) == 1:
        (q,) = qubits
        new_idx = f"((S.idx & ~(1 << {q})) | (U.row << {q}))"
        col = f"((S.idx >> {q}) & 1)"
        arity = 1
    else:
        q0, q1 = qubits
        new_idx = (f"((S.idx & ~((1 << {q0}) | (1 << {q1}))) | ((U.row & 1) << {q0}) "
                   f"| (((U.row >> 1) & 1) << {q1}))")
        col = f"(((S.idx >> {q0}) & 1) | (((S.idx >> {q1}) & 1) << 1))"
        arity = 2
    return (f"INSERT INTO state(version, idx, real, imag) "
            f"SELECT {v + 1}, {new_idx}, SUM(U.real * S.real - U.imag * S.imag), "
            f"SUM(U.real * S.imag + U.imag * S.real) FROM state AS S JOIN gate_matrix AS U "
            f"ON U.gate_name = '{key}' AND U.arity = {arity} AND U.col = {col} "
            f"WHERE S.version = {v} GROUP BY {new_idx};")


def run_circuit(circuit_dict: dict, return_row_counts: bool = False):
    """Run the circuit in SQLite; returns the dense complex128 state (and row counts)."""
    con = _connect()
    n = circuit_dict["number_of_qubits"]
    con.execute("INSERT INTO state VALUES (0, 0, 1.0, 0.0)")
    keyed = []
    for i, entry in enumerate(circuit_dict["gates"]):
        name, params, qubits = dense_oracle.decode_gate(entry)
        key = name if not params else f"{name}#{i}"
        _register(con, key, dense_oracle.gate_matrix(name, params))
        keyed.append((key, qubits))
    con.commit()
    for v, (key, qubits) in enumerate(keyed):
        con.execute(_gate_sql(key, qubits, v))
        con.commit()
    psi = np.zeros(1 << n, dtype=np.complex128)
    for idx, re, im in con.execute("SELECT idx, real, imag FROM state WHERE version = ?",
                                   (len(keyed),)):
        psi[idx] = complex(re, im)
    counts = [c for (c,) in con.execute(
        "SELECT COUNT(*) FROM state GROUP BY version ORDER BY version")]
    con.close()
    return (psi, counts) if return_row_counts else psi
