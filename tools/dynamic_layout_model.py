#!/usr/bin/env python3
"""Offline (no GPU): what would layouts that CHANGE between passes be worth?  A pass may store its tile with the tile's
qubits permuted among the tile's own index bits for free (lay_out), so qubits could migrate to cheap positions over the
passes.  Annealing over (initial placement + trades of two tile members after a pass) under the tile-cost model, against the
static placement the engine uses.
    python tools/dynamic_layout_model.py N [rand|clifft]"""
import sys, time
import numpy as np
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from quantum_simulations_amd.circuits import random_1q_cx_circuit, random_clifford_t_circuit
from quantum_simulations_amd.circuit.io import levelize, validate_circuit_dict
from quantum_simulations_amd.circuit.fusion import batch_levels
from quantum_simulations_amd.runner.engine import _planned_tile_masks
from quantum_simulations_amd.runner import tile_layout
n = int(sys.argv[1]); fam = sys.argv[2] if len(sys.argv) > 2 else "rand"
cd = random_1q_cx_circuit(n, depth=40) if fam == "rand" else random_clifford_t_circuit(n, depth=60)
ops = [p["local_ops"] for p in batch_levels(levelize(validate_circuit_dict(cd)), n)][0]
masks = _planned_tile_masks(n, ops)
tiles = [[b for b in range(3, n) if (int(m) >> b) & 1] for m in masks]
print(len(tiles), "tiles")
model = tile_layout.model_for(n)
nb = model["top"] - 3 + 1
bit, pair = model["bit"], model["pair"]
sym = pair + pair.T
def tcost(positions):
    idx = np.minimum(np.asarray(positions), model["top"]) - 3
    return model["c0"] + bit[idx].sum() + sym[np.ix_(idx, idx)].sum() / 2
def total(pos0, trades):
    pos = pos0.copy(); c = 0.0
    by_pass = {}
    for (p, a, b) in trades: by_pass.setdefault(p, []).append((a, b))
    for t, T in enumerate(tiles):
        c += tcost(pos[T])
        for a, b in by_pass.get(t, []):
            pos[a], pos[b] = pos[b], pos[a]
    return c
rng = np.random.default_rng(1)
def anneal(dynamic, iters, seed):
    rng = np.random.default_rng(seed)
    pos0 = np.arange(n); trades = []
    cur = total(pos0, trades); best = cur; T0 = 0.02 * cur / len(tiles)
    for it in range(iters):
        T = T0 * (1 - it / iters) + 1e-6
        kind = rng.random()
        if not dynamic or kind < 0.4:
            a, b = rng.choice(np.arange(3, n), size=2, replace=False)
            pos0[a], pos0[b] = pos0[b], pos0[a]
            new = total(pos0, trades)
            if new < cur or rng.random() < np.exp((cur - new) / T): cur = new
            else: pos0[a], pos0[b] = pos0[b], pos0[a]
        elif kind < 0.8 or not trades:
            p = int(rng.integers(len(tiles) - 1)); a, b = rng.choice(tiles[p], size=2, replace=False)
            trades.append((p, int(a), int(b)))
            new = total(pos0, trades)
            if new < cur or rng.random() < np.exp((cur - new) / T): cur = new
            else: trades.pop()
        else:
            i = int(rng.integers(len(trades))); tr = trades.pop(i)
            new = total(pos0, trades)
            if new < cur or rng.random() < np.exp((cur - new) / T): cur = new
            else: trades.insert(i, tr)
        best = min(best, cur)
    return best, len(trades)
ident = total(np.arange(n), [])
t = time.time()
s = min(anneal(False, 6000, sd)[0] for sd in range(3))
print(f"identity {ident:.2f} ms, static annealed {s:.2f} ({time.time()-t:.0f} s)")
t = time.time()
d = [anneal(True, 20000, sd) for sd in range(3)]
print(f"dynamic annealed {min(x[0] for x in d):.2f} trades {[x[1] for x in d]} ({time.time()-t:.0f} s)")
