#!/usr/bin/env python3
"""Offline (no GPU): how many passes if the three LINE-bit qubits could change from pass to pass?  (round 4)
A pass may permute the eleven qubits of its tile among the tile's eleven index bits when it stores (a bit permutation
inside the tile), so the next pass's line qubits can be ANY three of the current tile's eleven.  Emulated with the
library's planner: plan the remaining ops, take the first pass, decide its members with the pass builder's own
admissibility rule (restated below, record budget ignored), choose the next three line qubits among the tile's eleven by
one-step look-ahead (the triple whose next pass holds the most ops), relabel, repeat.
    python tools/dynamic_line_probe.py [n] [seed ...]"""
import ctypes as C
import itertools
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from quantum_simulations_amd import _lib  # noqa: E402
from quantum_simulations_amd.circuit.fusion import batch_levels  # noqa: E402
from quantum_simulations_amd.circuit.io import levelize, validate_circuit_dict  # noqa: E402
from quantum_simulations_amd.circuits import random_1q_cx_circuit  # noqa: E402
from quantum_simulations_amd.kernel.device import pack_ops  # noqa: E402

lib = _lib.load()
LOW = 3


def classify(qs, U):
    """(all qubits, general targets, X-type targets) as the pass builder sees an op (csrc/tile_planner.h classify_op)"""
    U = np.asarray(U)
    if len(qs) == 1:
        diag = U[0, 1] == 0 and U[1, 0] == 0
        if diag and U[0, 0] == 1:
            return set(qs), set(), set()
        xlike = U[0, 0] == 0 and U[1, 1] == 0 and U[0, 1] == 1 and U[1, 0] == 1
        return set(qs), set(qs), set(qs) if xlike else set()
    if not np.any(U - np.diag(np.diag(U))):
        return set(qs), set(), set()
    I2 = np.eye(2)
    for ctrl, tgt, blk in ((0, 1, U[2:, 2:]), (1, 0, U[np.ix_([1, 3], [1, 3])])):
        rest = U[:2, :2] if ctrl == 0 else U[np.ix_([0, 2], [0, 2])]
        off = np.delete(np.delete(U, [2, 3] if ctrl == 0 else [1, 3], 0), [0, 1] if ctrl == 0 else [0, 2], 1)
        if np.array_equal(rest, I2) and not np.any(off):
            xlike = blk[0, 0] == 0 and blk[1, 1] == 0 and blk[0, 1] == 1 and blk[1, 0] == 1
            return set(qs), {qs[tgt]}, {qs[tgt]} if xlike else set()
    return set(qs), set(qs), set()


def members_of(ops, tile, nmax=128):
    """indices of the ops a pass with tile qubits `tile` (line qubits included) holds: the rule of plan_fused::holds"""
    bt, bd, bx = set(), set(), set()
    out = []
    for i, (qm, tm, xm) in enumerate(ops):
        ok = not (qm & bt) and not (tm & bd) and not ((qm - xm) & bx) and tm <= tile
        if ok:
            out.append(i)
            if len(out) >= nmax:
                break
        else:
            bt |= tm - xm
            bx |= xm
            bd |= qm - tm
    return out


def first_pass(n, ops_list, label):
    """(tile as logical qubits, planned pass count) of the first pass the library plans for the ops under `label` (qubit -> bit)"""
    nq, qubits, mats = pack_ops([([label[q] for q in qs], U) for qs, U in ops_list])
    count = C.c_int32()
    args = (n, len(nq), nq.ctypes.data_as(C.c_void_p), qubits.ctypes.data_as(C.c_void_p), mats.ctypes.data_as(C.c_void_p))
    if len(nq) < 2:
        return set(range(n)), 1
    _lib.check(lib.qsim_plan_ops(*args, None, 0, C.byref(count)))
    img = np.zeros((count.value, 4096), dtype=np.uint8)
    _lib.check(lib.qsim_plan_ops(*args, img.ctypes.data_as(C.c_void_p), img.nbytes, C.byref(count)))
    T = int(img[0, 12:16].view("<i4")[0])
    bits = {int(b) for b in img[0, 16:16 + T - 3]} | {0, 1, 2}
    inv = {b: q for q, b in label.items()}
    return {inv[b] for b in bits}, count.value


def label_with_line(n, line):
    label, rest = {}, [q for q in range(n) if q not in line]
    for i, q in enumerate(line):
        label[q] = i
    for i, q in enumerate(rest):
        label[q] = LOW + i
    return label


def run(n, seed, lookahead_triples=True):
    cd = validate_circuit_dict(random_1q_cx_circuit(n, depth=40, seed=seed))
    ops_list = [op for p in batch_levels(levelize(cd), n) for op in p["local_ops"]]
    static = first_pass(n, ops_list, {q: q for q in range(n)})[1]
    line = [0, 1, 2]
    passes = 0
    t0 = time.time()
    while ops_list:
        label = label_with_line(n, line)
        tile, _ = first_pass(n, ops_list, label)
        cls = [classify(qs, U) for qs, U in ops_list]
        done = set(members_of(cls, tile))
        assert done, "no progress"
        ops_list = [op for i, op in enumerate(ops_list) if i not in done]
        passes += 1
        if not ops_list:
            break
        # next line qubits: any three of this tile's eleven; one-step look-ahead on the ops the next pass would hold
        cls = [classify(qs, U) for qs, U in ops_list]
        best = None
        for trip in itertools.combinations(sorted(tile), 3):
            lab = label_with_line(n, list(trip))
            t2, rem = first_pass(n, ops_list, lab)
            held = len(members_of(cls, t2))
            key = (held, -rem)
            if best is None or key > best[0]:
                best = (key, list(trip))
        line = best[1]
    return static, passes, time.time() - t0


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 28
    seeds = [int(s) for s in sys.argv[2:]] or [20260228]
    for seed in seeds:
        static, dyn, dt = run(n, seed)
        print(f"n={n} seed {seed}: static line qubits 0 1 2: {static} passes; line qubits re-chosen from every tile: {dyn} passes ({dt:.0f} s)", flush=True)
