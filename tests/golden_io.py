"""Readers for the committed golden fixtures (tests/golden/*, made by make_golden.py)."""
from __future__ import annotations

import json
from functools import lru_cache
from pathlib import Path

import numpy as np

GOLDEN = Path(__file__).resolve().parent / "golden"


def c128(pairs, shape=None) -> np.ndarray:
    a = np.array([complex(re, im) for re, im in pairs], dtype=np.complex128)
    return a.reshape(shape) if shape is not None else a


def circuit_from_json(doc: dict) -> dict:
    gates = []
    for g in doc["gates"]:
        e = {"qubits": list(g["qubits"]), "gate": g["gate"]}
        if "params" in g:
            p = {}
            for k, v in g["params"].items():
                if isinstance(v, dict) and "__c128__" in v:
                    p[k] = c128(v["__c128__"], v["shape"])
                else:
                    p[k] = v
            e["params"] = p
        gates.append(e)
    return {"number_of_qubits": doc["number_of_qubits"], "gates": gates}


def ops_from_json(items) -> list:
    return [(list(o["qubits"]), c128(o["U"], (o["dim"], o["dim"]))) for o in items]


@lru_cache(maxsize=None)
def npz(name: str):
    with np.load(GOLDEN / name, allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


@lru_cache(maxsize=None)
def jdoc(name: str):
    with open(GOLDEN / name) as f:
        return json.load(f)


def golden_circuits() -> dict:
    return {k: circuit_from_json(v) for k, v in jdoc("circuits.json").items()}
