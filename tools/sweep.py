#!/usr/bin/env python3
"""Per-target-qubit timing sweep (BASELINE config 3): H / T / CNOT on every target of an
n-qubit random state, HIP-event timed, reported as ms, algorithmic GB/s and fraction of
the 8 TB/s HBM peak.   python tools/sweep.py [n] [reps]"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from quantum_simulations_amd.kernel import gates as gt  # noqa: E402
from quantum_simulations_amd.kernel.device import DeviceChunk  # noqa: E402

PEAK = 8.0e12


def timed(dev, fn, reps):
    fn()
    dev.sync()
    ts = []
    for _ in range(reps):
        dev.time_begin()
        fn()
        ts.append(dev.time_end())
    return float(np.median(ts)), float(np.min(ts))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    only = set(sys.argv[3].split(";")) if len(sys.argv) > 3 else None
    dev = DeviceChunk.empty(n)
    dev.init_random(30)
    N = 1 << n
    H, T, CX, U2 = gt.H(), gt.T(), gt.CNOT(), None
    rng = np.random.default_rng(0)
    z = rng.standard_normal((4, 4)) + 1j * rng.standard_normal((4, 4))
    U2, _ = np.linalg.qr(z)
    rows = []
    want = lambda name: only is None or name in only
    for q in range(n if want("H") else 0):
        med, mn = timed(dev, lambda: dev.apply_1q(q, H), reps)
        rows.append(("H", q, med, mn, 32 * N))
    for q in range(n if want("T") else 0):
        med, mn = timed(dev, lambda: dev.apply_1q(q, T), reps)
        rows.append(("T", q, med, mn, 16 * N))
    for q in range(1, n if want("CX(0,q)") else 0):
        med, mn = timed(dev, lambda: dev.apply_2q(0, q, CX), reps)
        rows.append(("CX(0,q)", q, med, mn, 16 * N))
    for q in range(0, n - 1 if want("CX(q,q+1)") else 0):
        med, mn = timed(dev, lambda: dev.apply_2q(q, q + 1, CX), reps)
        rows.append(("CX(q,q+1)", q, med, mn, 16 * N))
    for q in range(0, n - 1 if want("U4(q+1,q)") else 0, 3):
        med, mn = timed(dev, lambda: dev.apply_2q(q + 1, q, U2), reps)
        rows.append(("U4(q+1,q)", q, med, mn, 32 * N))
    print(f"n={n} reps={reps} norm2={dev.norm2():.15f}")
    print(f"{'gate':<10} {'q':>3} {'med ms':>9} {'min ms':>9} {'GB/s':>9} {'frac8T':>7}")
    for name, q, med, mn, nbytes in rows:
        gbs = nbytes / (med * 1e-3) / 1e9
        print(f"{name:<10} {q:>3} {med:9.3f} {mn:9.3f} {gbs:9.1f} {gbs * 1e9 / PEAK:7.3f}")
    for name in ("H", "T", "CX(0,q)", "CX(q,q+1)", "U4(q+1,q)"):
        if not any(r[0] == name for r in rows):
            continue
        fr = [nb / (m * 1e-3) / PEAK for nm, _, m, _, nb in rows if nm == name]
        print(f"summary {name:<10} min frac {min(fr):.3f} median {np.median(fr):.3f} max {max(fr):.3f}")


if __name__ == "__main__":
    main()
