#!/usr/bin/env python3
"""Which of the minimum-pass layouts is fastest, and what predicts it?  (round 4)  For the bench circuit: every candidate
choice of line-bit qubits with the fewest passes among K, each with its annealed layout: model cost, gates that target a
line bit, measured step time.      python tools/layout_finalists_probe.py [n] [K]"""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from quantum_simulations_amd.circuit.fusion import batch_levels  # noqa: E402
from quantum_simulations_amd.circuit.io import levelize, validate_circuit_dict  # noqa: E402
from quantum_simulations_amd.circuits import random_1q_cx_circuit  # noqa: E402
from quantum_simulations_amd.kernel.device import DeviceChunk, pack_ops  # noqa: E402
from quantum_simulations_amd.runner import engine as eng_mod, tile_layout  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 28
K = int(sys.argv[2]) if len(sys.argv) > 2 else 512
cd = validate_circuit_dict(random_1q_cx_circuit(n, depth=40))
batches = [p["local_ops"] for p in batch_levels(levelize(cd), n)]
rng = np.random.default_rng(20260504)
layouts = np.tile(np.arange(n, dtype=np.int32), (K + 1, 1))
for row in layouts[1:]:
    for bit, q in enumerate(int(x) for x in rng.choice(n, size=3, replace=False)):
        j = int(np.flatnonzero(row == bit)[0])
        row[j], row[q] = row[q], bit
counts = eng_mod._count_passes(n, batches, layouts, 16)
best = int(counts.min())
print(f"n={n}: identity {counts[0]} passes, best {best}, histogram {dict(zip(*np.unique(counts, return_counts=True)))}", flush=True)
state = DeviceChunk.empty(n)
rows = []
cands = [0] + [int(i) for i in np.flatnonzero(counts == best)][:10] + [int(i) for i in np.flatnonzero(counts == best + 1)][:3]
for f in cands:
    first = [int(x) for x in layouts[f]]
    moved = [[([first[q] for q in qs], U) for qs, U in ops] for ops in batches]
    masks = [eng_mod._planned_tile_masks(n, ops) for ops in moved]
    tiles = [[b for b in range(3, n) if (int(m) >> b) & 1] for ms in masks for m in ms]
    second, c0, c1 = min((tile_layout.choose_layout(tiles, n, seed=s) for s in range(1, 9)), key=lambda r: r[2])
    l2p = [second[first[q]] for q in range(n)]
    fmasks = [np.array([sum(1 << second[b] for b in range(n) if (int(m) >> b) & 1) for m in ms], dtype=np.uint64) for ms in masks]
    packed = [pack_ops([([l2p[q] for q in qs], U) for qs, U in ops]) for ops in batches]
    line_targets = sum(1 for ops in batches for qs, U in ops if l2p[qs[-1]] < 3 or (len(qs) == 2 and l2p[qs[0]] < 3 and not np.allclose(U, np.diag(np.diag(U)))))
    line_gates = sum(1 for ops in batches for qs, U in ops if any(l2p[q] < 3 for q in qs))
    best_ms = 1e9
    for rep in range(2):
        state.init_zero(True)
        passes = sum(state.apply_ops_tiled(p, m) for p, m in zip(packed, fmasks))
        state.sync()
        t1 = time.perf_counter()
        for _ in range(5):
            for p, m in zip(packed, fmasks):
                state.apply_ops_tiled(p, m)
        state.sync()
        best_ms = min(best_ms, (time.perf_counter() - t1) / 5 * 1e3)
    print(f"cand {f:4d}: line qubits {[q for q in range(n) if first[q] < 3]}  {passes} passes  model {c1:7.2f} ms  gates on line bits {line_gates:3d} (targets {line_targets:3d})  "
          f"measured {best_ms:7.3f} ms = {best_ms / passes:.4f} ms/pass", flush=True)
state.close()
