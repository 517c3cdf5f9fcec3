#!/usr/bin/env python3
"""Fit the tile-bit cost model of runner/tile_layout.py: gate-less pass time of a tile-bit set = c0 + per-bit + pair terms
(ridge), from the sample files of tools/tile_bits_sample.py ("b0 ... b7 ms" per line).
    python tools/fit_tile_cost_model.py 28=profiles/r04k_tile_bits_samples_28q.txt 30=profiles/r04k_tile_bits_samples_30q.txt
writes quantum_simulations_amd/runner/tile_cost_model.json and prints the held-out error of every fit."""
import itertools
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
LOW = 3


def fit(path: Path, n: int, lam: float = 10.0) -> dict:
    rows = [ln.split() for ln in open(path) if ln.strip() and not ln.startswith("#")]
    tiles = [[int(x) for x in r[:8]] for r in rows]
    y = np.array([float(r[8]) for r in rows])
    top = n - 1
    nb = top - LOW + 1
    pairs = list(itertools.combinations(range(nb), 2))
    pidx = {p: i for i, p in enumerate(pairs)}

    def feat(t):
        f = np.zeros(1 + nb + len(pairs))
        f[0] = 1
        idx = sorted(b - LOW for b in t)
        f[1 + np.array(idx)] = 1
        for a, b in itertools.combinations(idx, 2):
            f[1 + nb + pidx[(a, b)]] = 1
        return f
    X = np.array([feat(t) for t in tiles])

    def solve(Xs, ys):
        A = Xs.T @ Xs + lam * np.eye(Xs.shape[1])
        A[0, 0] -= lam
        return np.linalg.solve(A, Xs.T @ ys)
    rng = np.random.default_rng(0)
    perm = rng.permutation(len(y))
    cut = len(y) * 4 // 5
    w = solve(X[perm[:cut]], y[perm[:cut]])
    held = float(np.sqrt(np.mean((X[perm[cut:]] @ w - y[perm[cut:]]) ** 2)))
    w = solve(X, y)
    pair = np.zeros((nb, nb))
    for (a, b), i in pidx.items():
        pair[a, b] = w[1 + nb + i]
    print(f"{n} qubits: {len(y)} samples from {path.name}, mean {y.mean():.4f} ms, std {y.std():.4f}, held-out rms {held:.4f} "
          f"(R^2 {1 - held ** 2 / y.var():.2f})")
    return {"top": top, "c0": float(w[0]), "bit": [float(x) for x in w[1:1 + nb]], "pair": [[float(x) for x in row] for row in pair],
            "samples": len(y), "source": f"profiles/{path.name}", "held_out_rms_ms": held, "mean_ms": float(y.mean()), "std_ms": float(y.std())}


def main():
    models = {}
    for arg in sys.argv[1:]:
        n, path = arg.split("=")
        models[n] = fit(ROOT / path, int(n))
    out = ROOT / "quantum_simulations_amd" / "runner" / "tile_cost_model.json"
    out.write_text(json.dumps({"what": "gate-less fused-pass time by tile-bit set on MI355X: c0 + sum bit[b - 3] + sum pair[a - 3][b - 3] (a < b), "
                                       "ridge fit (tools/fit_tile_cost_model.py)", "low": LOW, "models": models}))
    print("wrote", out)


if __name__ == "__main__":
    main()
