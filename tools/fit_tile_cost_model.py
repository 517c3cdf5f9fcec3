#!/usr/bin/env python3
"""Fit the tile-bit cost model of runner/tile_layout.py: gate-less pass time of a tile-bit set = c0 + per-bit + pair terms
(ridge), from the sample files of tools/tile_bits_sample.py ("b0 ... b7 ms" per line).
    python tools/fit_tile_cost_model.py [--triples] [--lambda=L] 28=profiles/a.txt[,profiles/b.txt] 30=profiles/c.txt
writes quantum_simulations_amd/runner/tile_cost_model.json and prints the held-out error of every fit."""
import itertools
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
LOW = 3


def fit(paths, n: int, lam: float = 10.0, triples: bool = False) -> dict:
    rows = [ln.split() for path in paths for ln in open(path) if ln.strip() and not ln.startswith("#")]
    tiles = [[int(x) for x in r[:8]] for r in rows]
    y = np.array([float(r[8]) for r in rows])
    top = n - 1
    nb = top - LOW + 1
    pairs = list(itertools.combinations(range(nb), 2))
    pidx = {p: i for i, p in enumerate(pairs)}
    tris = list(itertools.combinations(range(nb), 3)) if triples else []
    tidx = {t: i for i, t in enumerate(tris)}

    def feat(t):
        f = np.zeros(1 + nb + len(pairs) + len(tris), dtype=np.float32)
        f[0] = 1
        idx = sorted(b - LOW for b in t)
        f[1 + np.array(idx)] = 1
        for a, b in itertools.combinations(idx, 2):
            f[1 + nb + pidx[(a, b)]] = 1
        if triples:
            for abc in itertools.combinations(idx, 3):
                f[1 + nb + len(pairs) + tidx[abc]] = 1
        return f
    X = np.array([feat(t) for t in tiles], dtype=np.float64)

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
    names = ", ".join(p.name for p in paths)
    print(f"{n} qubits: {len(y)} samples from {names}, mean {y.mean():.4f} ms, std {y.std():.4f}, held-out rms {held:.4f} "
          f"(R^2 {1 - held ** 2 / y.var():.2f}){' with third-order terms' if triples else ''}")
    doc = {"top": top, "c0": float(w[0]), "bit": [float(x) for x in w[1:1 + nb]], "pair": [[float(x) for x in row] for row in pair],
           "samples": len(y), "source": [f"profiles/{p.name}" for p in paths], "held_out_rms_ms": held, "mean_ms": float(y.mean()),
           "std_ms": float(y.std())}
    if triples:
        base = 1 + nb + len(pairs)
        doc["tri"] = [[a, b, c, float(w[base + i])] for (a, b, c), i in tidx.items()]
    return doc


def main():
    models = {}
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    triples = "--triples" in sys.argv
    lam = next((float(a.split("=")[1]) for a in sys.argv if a.startswith("--lambda=")), 30.0 if triples else 10.0)
    for arg in args:                                     # n=file[,file...]
        n, paths = arg.split("=")
        models[n] = fit([ROOT / p for p in paths.split(",")], int(n), lam=lam, triples=triples)
    out = ROOT / "quantum_simulations_amd" / "runner" / "tile_cost_model.json"
    out.write_text(json.dumps({"what": "gate-less fused-pass time by tile-bit set on MI355X: c0 + sum bit[b - 3] + sum pair[a - 3][b - 3] (a < b), "
                                       "ridge fit (tools/fit_tile_cost_model.py)", "low": LOW, "models": models}))
    print("wrote", out)


if __name__ == "__main__":
    main()
