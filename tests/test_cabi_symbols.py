"""CPU-side checks of the drop-in boundary: libqsim_hip.so loads, exports every symbol that
include/qsim_hip.h declares, the ctypes table covers exactly that set, and without a GPU the
library fails loudly (no silent CPU fallback).  No compute calls here."""
import ctypes
import re
from pathlib import Path

import pytest

from quantum_simulations_amd import _lib

ROOT = Path(__file__).resolve().parent.parent
HEADER = ROOT / "include" / "qsim_hip.h"


def _declared():
    text = re.sub(r"/\*.*?\*/", "", HEADER.read_text(), flags=re.S)
    return set(re.findall(r"\b(qsim_[a-z0-9_]+)\s*\(", text))


def test_header_declares_the_six_reference_entry_points():
    names = _declared()
    for fn in ("qsim_apply_1q", "qsim_apply_2q", "qsim_apply_1q_pair", "qsim_apply_2q_pair_qa_local",
               "qsim_apply_2q_pair_qb_local", "qsim_apply_2q_quad"):
        assert fn in names


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    declared = _declared()
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name), name


def test_no_torch_types_and_c_linkage():
    text = HEADER.read_text()
    assert 'extern "C"' in text and "torch" not in text.lower().replace("torch tensor", "")


def test_fails_loudly_without_gpu_or_library(monkeypatch, tmp_path):
    lib = _lib.load()
    n = ctypes.c_int(-1)
    rc = lib.qsim_device_count(ctypes.byref(n))
    if rc != 0:  # no GPU in this container: error code + message, and Python raises
        assert rc == _lib.QSIM_ERR_HIP and lib.qsim_last_error()
        from quantum_simulations_amd.kernel.device import DeviceChunk
        with pytest.raises((_lib.QsimHipError, MemoryError, ValueError)):
            DeviceChunk.zero_state(3)
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", tmp_path / "missing.so")
    with pytest.raises(_lib.QsimLibraryMissing):
        _lib.load()


def test_product_package_never_imports_oracle():
    pkg = ROOT / "quantum_simulations_amd"
    for path in pkg.rglob("*.py"):
        src = path.read_text()
        assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), path
