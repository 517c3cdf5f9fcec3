"""Qubit layout chosen for the DRAM pattern of the fused passes' tiles (round 4).

What a fused pass costs at 28-30 qubits is decided by WHICH index bits its tile holds (1.32-2.3 ms
for the same data volume; the gate-less time of a tile-bit set predicts the time of the pass WITH its gates at r = 0.83).
The pass builder cannot pay for a better pattern with extra passes (measured), but the engine is free to decide which
index bit a logical qubit lives on: the passes -- their tiles as SETS OF QUBITS -- stay exactly the same and only the
index bits under them change.  This module holds

  * a cost model of a tile-bit set: constant + per-bit terms + pair terms, a ridge fit to a few thousand gate-less passes
    over random tile-bit sets measured on MI355X (`tools/fit_tile_cost_model.py` <- `tools/tile_bits_sample.py`; the
    coefficients ship as `tile_cost_model.json`);
  * `choose_layout(tiles, n)`: simulated annealing over the assignment qubit -> index bit (bits 0-2, the 128-byte line,
    stay) minimising the model's total over the circuit's passes -- a second or so on the host, outside any timed region.

The reference has the same notion for another purpose: `atlas_stages` returns `log_to_phys` and `permute_state` undoes it
(wenbo_engine/circuit/staging.py:587-658); `SingleGpuEngine` tracks its layout the same way.
"""
from __future__ import annotations

import json
from pathlib import Path

import numpy as np

LOW = 3                                              # bits 0..2 = one 128-byte line: always tile bits, never moved
_MODEL_PATH = Path(__file__).resolve().parent / "tile_cost_model.json"
_models: dict | None = None


def _load_models() -> dict:
    global _models
    if _models is None:
        doc = json.loads(_MODEL_PATH.read_text())
        _models = {}
        for k, m in doc["models"].items():
            nb = int(m["top"]) - LOW + 1
            tri = None
            if m.get("tri"):                         # third-order terms [[a, b, c, w], ...] (a < b < c, bit - 3)
                tri = np.zeros((nb, nb, nb), dtype=np.float64)
                for a, b, c, w in m["tri"]:
                    tri[int(a), int(b), int(c)] = w
            _models[int(k)] = {"c0": m["c0"], "bit": np.asarray(m["bit"], dtype=np.float64),
                               "pair": np.asarray(m["pair"], dtype=np.float64), "tri": tri, "top": int(m["top"])}
    return _models


def model_for(n: int) -> dict:
    """The fit made at the state size nearest to n (an index bit is an address bit whatever n is; bits above the
    fitted range are priced like the highest fitted one)."""
    models = _load_models()
    key = min(models, key=lambda k: (abs(k - n), k))
    return models[key]


def tile_cost(model: dict, bits) -> float:
    """Predicted milliseconds (at the model's own state size) of a pass whose tile holds the index bits `bits` (>= 3)."""
    idx = sorted(min(int(b), model["top"]) - LOW for b in bits)
    c = model["c0"] + float(model["bit"][idx].sum())
    pair, tri = model["pair"], model.get("tri")
    for i, a in enumerate(idx):
        for j in range(i + 1, len(idx)):
            b = idx[j]
            if a == b:
                continue
            c += float(pair[a, b])
            if tri is not None:
                for d in idx[j + 1:]:
                    if d != b:
                        c += float(tri[a, b, d])
    return c


def choose_layout(tiles: list, n: int, seed: int = 1, sweeps: int = 200) -> tuple:
    """tiles: the high tile bits (logical qubits >= 3) of every pass of the circuit.  Returns (l2p, cost_identity,
    cost_chosen) in the model's milliseconds: logical qubit q lives on index bit l2p[q].  The annealing itself is host
    code of the library (qsim_choose_layout: ~10 ms for the bench circuits)."""
    import ctypes as C

    from quantum_simulations_amd import _lib
    model = model_for(n)
    if n < LOW + 2 or not tiles:
        return list(range(n)), 0.0, 0.0
    masks = np.array([sum(1 << int(q) for q in t if q >= LOW) for t in tiles], dtype=np.uint64)
    bit = np.ascontiguousarray(model["bit"], dtype=np.float64)
    pair = np.ascontiguousarray(model["pair"], dtype=np.float64)
    tri = None if model.get("tri") is None else np.ascontiguousarray(model["tri"], dtype=np.float64)
    out = np.zeros(n, dtype=np.int32)
    c0, c1 = C.c_double(), C.c_double()
    _lib.check(_lib.load().qsim_choose_layout(n, len(masks), masks.ctypes.data_as(C.c_void_p), model["top"],
                                              bit.ctypes.data_as(C.c_void_p), pair.ctypes.data_as(C.c_void_p),
                                              None if tri is None else tri.ctypes.data_as(C.c_void_p), seed, sweeps,
                                              out.ctypes.data_as(C.c_void_p), C.byref(c0), C.byref(c1)))
    base = model["c0"] * len(tiles)
    return [int(x) for x in out], c0.value + base, c1.value + base
