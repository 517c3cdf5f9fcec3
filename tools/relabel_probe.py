#!/usr/bin/env python3
"""Does a static relabelling of the qubits (logical qubit q lives on index bit pi(q) for the whole run) chosen from a
memory-pattern model of the tile bits make the fused passes faster?  (round 4)
  * model: ridge fit of per-bit + pair terms to profiles/r02z_tile_bits_samples.txt (2500 gate-less passes by tile-bit set)
  * tiles of the circuit: planned on the CPU (qsim_plan_ops), identity labels
  * pi: simulated annealing on the model's predicted total over the circuit's passes
  * measurement: the circuit with identity labels and with pi, same device, alternating
    python tools/relabel_probe.py [n] [seed ...]"""
import itertools
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from quantum_simulations_amd.circuit.fusion import batch_levels  # noqa: E402
from quantum_simulations_amd.circuit.io import levelize, validate_circuit_dict  # noqa: E402
from quantum_simulations_amd.circuits import random_1q_cx_circuit  # noqa: E402
from quantum_simulations_amd.kernel.device import pack_ops  # noqa: E402
from quantum_simulations_amd.runner.engine import make_engine  # noqa: E402
from tests import tile_interpreter as ti  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 28
seeds = [int(s) for s in sys.argv[2:]] or [20260228, 1, 2, 3]
LOW = 3
bits = list(range(LOW, 28))
nb = len(bits)
pairs = list(itertools.combinations(range(nb), 2))
pidx = {p: i for i, p in enumerate(pairs)}


def feat(tile):
    f = np.zeros(1 + nb + len(pairs))
    f[0] = 1
    idx = sorted(min(b, 27) - LOW for b in tile)          # (bits above 27: treated like bit 27)
    for i in idx:
        f[1 + i] = 1
    for a, b in itertools.combinations(idx, 2):
        if a != b:
            f[1 + nb + pidx[(a, b)]] = 1
    return f


rows = [ln.split() for ln in open(ROOT / "profiles" / "r02z_tile_bits_samples.txt") if ln.strip() and not ln.startswith("#")]
X = np.array([feat([int(x) for x in r[:8]]) for r in rows])
y = np.array([float(r[8]) for r in rows])
A = X.T @ X + 10 * np.eye(X.shape[1])
A[0, 0] -= 10
w = np.linalg.solve(A, X.T @ y)


def tiles_of(cd):
    out = []
    for p in batch_levels(levelize(validate_circuit_dict(cd)), n):
        for img in ti.plan(n, p["local_ops"]):
            out.append([int(b) for b in img["h"][:8]])
    return out


def relabel(cd, pi):
    return {"number_of_qubits": cd["number_of_qubits"],
            "gates": [dict(g, qubits=[pi[q] for q in g["qubits"]]) for g in cd["gates"]]}


def optimise(tiles, rng, iters=30000):
    pos = list(range(LOW, n))
    pi = {b: b for b in pos}

    def total(pi):
        return sum(feat([pi[b] for b in t]) @ w for t in tiles)
    cur = best = total(pi)
    bestpi = dict(pi)
    T = 0.05
    for _ in range(iters):
        a, b = (int(x) for x in rng.choice(pos, 2, replace=False))
        pi[a], pi[b] = pi[b], pi[a]
        new = total(pi)
        if new < cur or rng.random() < np.exp((cur - new) / T):
            cur = new
        else:
            pi[a], pi[b] = pi[b], pi[a]
        if cur < best:
            best, bestpi = cur, dict(pi)
        T = max(0.002, T * 0.9997)
    return bestpi, total({b: b for b in pos}), best


from quantum_simulations_amd.kernel.device import DeviceChunk  # noqa: E402

state = DeviceChunk.empty(n)
rng = np.random.default_rng(5)
tot = {"identity": [0.0, 0], "relabelled": [0.0, 0], "relabelled, same tiles": [0.0, 0]}


def batches_of(cd):
    return [p["local_ops"] for p in batch_levels(levelize(validate_circuit_dict(cd)), n)]


def masks_of(ops_list):
    """tile masks per batch, from the CPU planner"""
    return [np.array([sum(1 << int(b) for b in img["h"][:8]) for img in ti.plan(n, ops)], dtype=np.uint64) for ops in ops_list]


for seed in seeds:
    cd = random_1q_cx_circuit(n, depth=40, seed=seed)
    tiles = tiles_of(cd)
    t0 = time.time()
    pi, pred0, pred1 = optimise(tiles, rng)
    full = {q: q for q in range(LOW)}
    full.update(pi)
    cd2 = relabel(validate_circuit_dict(cd), full)
    print(f"seed {seed}: {len(tiles)} passes, model {pred0:.2f} ms -> {pred1:.2f} ms for the same tiles relabelled (search {time.time() - t0:.1f} s)", flush=True)
    ident_batches = batches_of(cd)
    relab_batches = batches_of(cd2)
    ident_masks = masks_of(ident_batches)
    # the SAME tiles on the new bits: map every mask of the identity plan through pi
    forced = [np.array([sum(1 << full[b] for b in range(n) if (int(m) >> b) & 1) for m in ms], dtype=np.uint64) for ms in ident_masks]
    variants = {"identity": [(pack_ops(o), None) for o in ident_batches],
                "relabelled": [(pack_ops(o), None) for o in relab_batches],
                "relabelled, same tiles": [(pack_ops(o), f) for o, f in zip(relab_batches, forced)]}
    res = {}
    for rep in range(2):
        for name, plan in variants.items():
            state.init_zero(True)

            def step():
                p = 0
                for packed, f in plan:
                    p += state.apply_ops(packed) if f is None else state.apply_ops_tiled(packed, f)
                return p
            passes = step()
            state.sync()
            t1 = time.perf_counter()
            for _ in range(5):
                step()
            state.sync()
            ms = (time.perf_counter() - t1) / 5 * 1e3
            res.setdefault(name, []).append((ms, passes))
            assert abs(state.norm2() - 1.0) < 1e-9
    for name, v in res.items():
        ms = min(x[0] for x in v)
        p = v[0][1]
        tot[name][0] += ms
        tot[name][1] += p
        print(f"   {name:24s}: {ms:7.3f} ms  {p} passes  {ms / p:.4f} ms/pass", flush=True)
for name, (ms, p) in tot.items():
    print(f"total {name:24s}: {ms:.3f} ms  {p} passes  {ms / p:.4f} ms/pass")
state.close()
