"""Step log of a checkpointed run: the reference's `wal.json` document.

Same file and field layout as wenbo_engine/wal/wal.py:25-93 -- {"circuit_hash", "committed_buf",
"done_steps"} rewritten atomically (tmp + fsync + rename) -- and the same circuit identity
(wal.py:17-22: sha256 of the validated circuit dict, first 16 hex digits), so either side can
resume a run the other one checkpointed.  The GPU runner commits every `checkpoint_every` steps
instead of every step (the state lives in HBM; a checkpoint is a full download).
"""
from __future__ import annotations

import hashlib
import json
import os
from pathlib import Path

from quantum_simulations_amd.circuit.io import validate_circuit_dict


def circuit_hash(circuit_dict: dict) -> str:
    canonical = json.dumps(validate_circuit_dict(circuit_dict), sort_keys=True, default=str)
    return hashlib.sha256(canonical.encode()).hexdigest()[:16]


class WAL:
    def __init__(self, path: str | Path, circuit_dict: dict | None = None):
        self.path = Path(path)
        self.path.parent.mkdir(parents=True, exist_ok=True)
        mine = circuit_hash(circuit_dict) if circuit_dict else None
        if self.path.exists():
            with open(self.path) as f:
                self._doc = json.load(f)
            theirs = self._doc.get("circuit_hash")
            if mine and theirs and mine != theirs:
                raise ValueError(f"WAL circuit hash mismatch — different circuit? WAL={theirs} vs new={mine}")
        else:
            self._doc = {"circuit_hash": mine or "", "committed_buf": "a", "done_steps": 0}
            self._store()

    def _store(self) -> None:
        tmp = self.path.with_suffix(".tmp")
        with open(tmp, "w") as f:
            f.write(json.dumps(self._doc, indent=2))
            f.flush()
            os.fsync(f.fileno())
        os.replace(tmp, self.path)

    @property
    def committed_buf(self) -> str:
        return self._doc.get("committed_buf", "a")

    @property
    def done_steps(self) -> int:
        return self._doc.get("done_steps", 0)

    def commit_step(self, step_idx: int, new_buf: str) -> None:
        self._doc["committed_buf"] = new_buf
        self._doc["done_steps"] = step_idx + 1
        self._store()

    def close(self) -> None:
        pass
