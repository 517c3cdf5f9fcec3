"""Export / import a state in the reference's on-disk chunk format.

The reference keeps every state as `<buffer>/chunks/chunk_%06d.bin` (raw complex64) plus
`<buffer>/manifest.json` (wenbo_engine/storage/block_store.py:11-65, manifest.py:19-64,
docs/storage_spec.md) and, after staging, `<work_dir>/qubit_mapping.json`
(runner/single_node.py:129-134).  This build keeps states in HBM; these helpers write a
finished run in that format, so the reference's own `collect_state(buf_path, ...)` can read a
GPU result, and read such a directory back into HBM.  Checkpoints of the GPU runner
(runner/single_node.run(checkpoint_every=...)) use the same layout with `"dtype": "complex128"`
so a resumed run of THIS build loses nothing (the reference's `Manifest.validate` accepts complex64
only: checkpoints it should read must be written with checkpoint_dtype="complex64"); complex64 stays
the default for exports.
"""
from __future__ import annotations

import json
import os
import time
from pathlib import Path

import numpy as np

DTYPE = np.complex64
DTYPES = {"complex64": np.complex64, "complex128": np.complex128}


def chunk_filename(idx: int) -> str:
    return f"chunk_{idx:06d}.bin"


def _replace_atomically(path: Path, payload: bytes) -> None:
    tmp = path.with_suffix(path.suffix + ".tmp") if path.suffix != ".bin" else path.with_suffix(".tmp")
    with open(tmp, "wb") as f:
        f.write(payload)
        f.flush()
        os.fsync(f.fileno())
    os.replace(tmp, path)


def write_state(directory: str | Path, state, chunk_size: int = 1 << 20,
                log_to_phys: list[int] | None = None, work_dir: str | Path | None = None,
                dtype: str = "complex64") -> Path:
    """Write `state` (a `DeviceChunk`, an `HbmStateBuffer` or a complex ndarray) as a reference
    buffer directory.  Amplitudes are rounded to complex64 like every reference chunk file unless
    `dtype="complex128"` (checkpoints)."""
    if dtype not in DTYPES:
        raise ValueError(f"unsupported dtype {dtype}")
    if hasattr(state, "state") and hasattr(state, "n_qubits"):       # HbmStateBuffer
        log_to_phys = log_to_phys if log_to_phys is not None else getattr(state, "log_to_phys", None)
        work_dir = work_dir if work_dir is not None else state.work_dir
        state = state.state
    total = len(state)
    n = total.bit_length() - 1
    if total != 1 << n:
        raise ValueError("state length must be a power of two")
    chunk_size = min(chunk_size, total)
    if total % chunk_size:
        raise ValueError("2^n_qubits must be divisible by chunk_size")
    d = Path(directory)
    (d / "chunks").mkdir(parents=True, exist_ok=True)
    names = []
    for c in range(total // chunk_size):
        if dtype == "complex64" and hasattr(state, "download_c64"):
            part = state.download_c64(c * chunk_size, chunk_size)     # rounded on the device: half the PCIe bytes
        elif hasattr(state, "download"):
            part = state.download(c * chunk_size, chunk_size)
        else:
            part = np.asarray(state[c * chunk_size:(c + 1) * chunk_size])
        name = chunk_filename(c)
        _replace_atomically(d / "chunks" / name, np.ascontiguousarray(part, dtype=DTYPES[dtype]).tobytes())
        names.append(name)
    manifest = {"n_qubits": n, "chunk_size": chunk_size, "n_chunks": len(names),
                "dtype": dtype, "chunks": names, "created": time.time()}
    _replace_atomically(d / "manifest.json", json.dumps(manifest, indent=2).encode())
    if log_to_phys and list(log_to_phys) != list(range(n)) and work_dir is not None:
        Path(work_dir).mkdir(parents=True, exist_ok=True)
        with open(Path(work_dir) / "qubit_mapping.json", "w") as f:
            json.dump(list(log_to_phys), f)
    return d


def read_manifest(directory: str | Path) -> dict:
    with open(Path(directory) / "manifest.json") as f:
        m = json.load(f)
    if m["chunk_size"] * m["n_chunks"] != 1 << m["n_qubits"]:
        raise ValueError(f"chunk_size*n_chunks={m['chunk_size'] * m['n_chunks']} != 2^n_qubits={1 << m['n_qubits']}")
    if len(m["chunks"]) != m["n_chunks"]:
        raise ValueError(f"chunk list length {len(m['chunks'])} != n_chunks {m['n_chunks']}")
    if m.get("dtype", "complex64") not in DTYPES:
        raise ValueError(f"unsupported dtype {m['dtype']}")
    return m


def read_state(directory: str | Path) -> np.ndarray:
    """A reference buffer directory -> complex128 vector (physical order, as stored)."""
    m = read_manifest(directory)
    dt = DTYPES[m.get("dtype", "complex64")]
    parts = [np.fromfile(str(Path(directory) / "chunks" / name), dtype=dt) for name in m["chunks"]]
    return np.concatenate(parts).astype(np.complex128)


def load_to_device(directory: str | Path, device: int = 0):
    """A reference buffer directory -> HBM-resident `DeviceChunk` (uploaded chunk by chunk)."""
    from quantum_simulations_amd.kernel.device import DeviceChunk
    m = read_manifest(directory)
    dev = DeviceChunk.empty(m["n_qubits"], device)
    dt = DTYPES[m.get("dtype", "complex64")]
    for c, name in enumerate(m["chunks"]):
        part = np.fromfile(str(Path(directory) / "chunks" / name), dtype=dt)
        if dt is np.complex64:
            dev.upload_c64(part, offset=c * m["chunk_size"])            # widened on the device
        else:
            dev.upload(part, offset=c * m["chunk_size"])
    return dev
