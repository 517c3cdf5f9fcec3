"""ctypes binding of libqsim_hip.so (C ABI declared in include/qsim_hip.h).

There is deliberately no fallback: if the shared library has not been built
(`python -c "import __graft_entry__ as g; g.build()"` or `make -C
quantum_simulations_amd/csrc`) loading raises `QsimLibraryMissing`.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

LIB_NAME = "libqsim_hip.so"
# QSIM_LIBRARY: another build of the same ABI (tools/ load the probe build libqsim_hip_probes.so this way;
# bench.py refuses to run with any QSIM_* variable set, so a timed number always comes from libqsim_hip.so)
LIB_PATH = Path(os.environ.get("QSIM_LIBRARY") or (Path(__file__).resolve().parent / LIB_NAME))

QSIM_OK = 0
QSIM_ERR_INVALID = -1
QSIM_ERR_NONLOCAL = -2
QSIM_ERR_HIP = -3
QSIM_ERR_NOMEM = -4


class QsimLibraryMissing(RuntimeError):
    pass


class QsimHipError(RuntimeError):
    pass


_P = C.c_void_p
_D = C.POINTER(C.c_double)
_I32 = C.POINTER(C.c_int32)

# name -> (restype, argtypes); every symbol include/qsim_hip.h declares
SIGNATURES = {
    "qsim_last_error": (C.c_char_p, []),
    "qsim_version": (C.c_int, []),
    "qsim_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "qsim_create": (C.c_int, [C.c_int, C.c_int, C.POINTER(_P)]),
    "qsim_create_view": (C.c_int, [_P, C.c_uint64, C.c_int, C.POINTER(_P)]),
    "qsim_wrap": (C.c_int, [C.c_int, _P, C.c_int, _P, C.POINTER(_P)]),
    "qsim_destroy": (C.c_int, [_P]),
    "qsim_n_local_qubits": (C.c_int, [_P]),
    "qsim_device_ptr": (_P, [_P]),
    "qsim_init_zero": (C.c_int, [_P, C.c_int]),
    "qsim_init_random": (C.c_int, [_P, C.c_uint64]),
    "qsim_upload": (C.c_int, [_P, _P, C.c_uint64, C.c_uint64]),
    "qsim_download": (C.c_int, [_P, _P, C.c_uint64, C.c_uint64]),
    "qsim_download_c64": (C.c_int, [_P, _P, C.c_uint64, C.c_uint64]),
    "qsim_upload_c64": (C.c_int, [_P, _P, C.c_uint64, C.c_uint64]),
    "qsim_copy": (C.c_int, [_P, _P]),
    "qsim_copy_variant": (C.c_int, [_P, _P, C.c_int]),
    "qsim_apply_1q": (C.c_int, [_P, C.c_int, _P]),
    "qsim_apply_2q": (C.c_int, [_P, C.c_int, C.c_int, _P]),
    "qsim_apply_fused_k": (C.c_int, [_P, C.c_int, _P, _P]),
    "qsim_apply_ops": (C.c_int, [_P, C.c_int, _P, _P, _P]),
    "qsim_apply_ops_unfused": (C.c_int, [_P, C.c_int, _P, _P, _P]),
    "qsim_plan_ops": (C.c_int, [C.c_int, C.c_int, _P, _P, _P, _P, C.c_uint64, _P]),
    "qsim_plan_ops_tiled": (C.c_int, [C.c_int, C.c_int, _P, _P, _P, C.c_int, _P, _P, C.c_uint64, _P]),
    "qsim_apply_ops_tiled": (C.c_int, [_P, C.c_int, _P, _P, _P, C.c_int, _P]),
    "qsim_choose_layout": (C.c_int, [C.c_int, C.c_int, _P, C.c_int, _P, _P, _P, C.c_uint64, C.c_int, _P, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "qsim_plan_count_layouts": (C.c_int, [C.c_int, C.c_int, _P, _P, _P, C.c_int, _P, _P, C.c_int]),
    "qsim_plan_peek_pass": (C.c_int, [C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64),
                                      C.POINTER(C.c_uint64), C.POINTER(C.c_int32), _P]),
    "qsim_last_pass_count": (C.c_int, [_P]),
    "qsim_plan_cache_clear": (C.c_int, []),
    "qsim_apply_ops_io": (C.c_int, [_P, C.c_int, _P, _P, _P, _P, C.POINTER(C.c_int)]),
    "qsim_apply_ops_io_part": (C.c_int, [_P, C.c_int]),
    "qsim_apply_ops_io_load": (C.c_int, [_P, C.c_int]),
    "qsim_apply_ops_io_source_parts": (C.c_int, [_P, _P, _P, _P]),
    "qsim_apply_ops_io_parts": (C.c_int, [_P, _P, _P, _P]),
    "qsim_apply_ops_io_own_slab": (C.c_int, [_P, _P]),
    "qsim_split_piece_count": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "qsim_apply_1q_pair": (C.c_int, [_P, _P, _P]),
    "qsim_apply_2q_pair_qa_local": (C.c_int, [_P, _P, C.c_int, _P]),
    "qsim_apply_2q_pair_qb_local": (C.c_int, [_P, _P, C.c_int, _P]),
    "qsim_apply_2q_quad": (C.c_int, [_P, _P, _P, _P, _P]),
    "qsim_pack_half": (C.c_int, [_P, C.c_int, C.c_int, _P]),
    "qsim_unpack_half": (C.c_int, [_P, C.c_int, C.c_int, _P]),
    "qsim_pack_bits": (C.c_int, [_P, C.c_int, _P, C.c_int, _P, C.c_uint64]),
    "qsim_unpack_bits": (C.c_int, [_P, C.c_int, _P, C.c_int, _P, C.c_uint64]),
    "qsim_pack_all": (C.c_int, [_P, C.c_int, _P, _P, C.c_int, C.c_int, C.c_int]),
    "qsim_unpack_all": (C.c_int, [_P, C.c_int, _P, _P, C.c_int, C.c_int, C.c_int]),
    "qsim_swap_global_local": (C.c_int, [_P, C.c_int, _P, _P, C.c_int]),
    "qsim_comm_get_unique_id": (C.c_int, [_P]),
    "qsim_comm_init": (C.c_int, [C.c_int, C.c_int, C.c_int, _P, C.POINTER(_P)]),
    "qsim_comm_destroy": (C.c_int, [_P]),
    "qsim_comm_rank": (C.c_int, [_P]),
    "qsim_comm_world": (C.c_int, [_P]),
    "qsim_comm_exchange": (C.c_int, [_P, C.c_int, _P, _P, _P, _P, _P, C.c_uint64]),
    "qsim_comm_exchange_bg": (C.c_int, [_P, C.c_int, _P, _P, _P, _P, _P, C.c_uint64, C.POINTER(C.c_uint32)]),
    "qsim_comm_wait": (C.c_int, [_P, _P, C.c_uint32]),
    "qsim_comm_join": (C.c_int, [_P, _P]),
    "qsim_comm_relayout": (C.c_int, [_P, _P, _P, _P, C.c_int, _P, _P, C.c_int]),
    "qsim_comm_relayout_plan": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, C.c_int, _P, _P, _P, _P, _P, _P]),
    "qsim_comm_relayout_loopback": (C.c_int, [_P, _P, _P, _P, C.c_int, _P, _P, C.c_int, C.c_int, C.c_int]),
    "qsim_apply_1q_pair_remote": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, _P]),
    "qsim_apply_2q_pair_qa_local_remote": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, _P]),
    "qsim_apply_2q_pair_qb_local_remote": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, _P]),
    "qsim_apply_2q_quad_remote": (C.c_int, [_P, _P, _P, _P, C.c_int, _P]),
    "qsim_comm_relayout_fused": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int, _P, _P, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]),
    "qsim_count_nonzero": (C.c_int, [_P, C.c_double, C.POINTER(C.c_uint64)]),
    "qsim_export_nonzero": (C.c_int, [_P, C.c_double, C.c_uint64, _P, _P, C.POINTER(C.c_uint64)]),
    "qsim_sync": (C.c_int, [_P]),
    "qsim_norm2": (C.c_int, [_P, C.POINTER(C.c_double)]),
    "qsim_max_abs_err_closed_form": (C.c_int, [_P, C.c_int, C.c_int, C.c_uint64,
                                               C.POINTER(C.c_double)]),
    "qsim_max_abs_err_closed_form_perm": (C.c_int, [_P, C.c_int, C.c_int, C.c_uint64, _P,
                                                    C.POINTER(C.c_double)]),
    "qsim_fingerprint": (C.c_int, [_P, C.c_int, C.c_uint64, _P, C.c_uint64, C.c_uint64, C.c_uint64, _P]),
    "qsim_time_begin": (C.c_int, [_P]),
    "qsim_time_end": (C.c_int, [_P, C.POINTER(C.c_float)]),
    "qsim_profile_begin": (C.c_int, [_P]),
    "qsim_profile_end": (C.c_int, [_P, C.c_int, C.POINTER(C.c_int), _P]),
}


class OpsIo(C.Structure):
    """qsim_ops_io (include/qsim_hip.h): buffers a re-layout is fused with at the ends of an op list."""
    _fields_ = [("struct_size", C.c_uint32),
                ("src", C.c_void_p), ("src_m", C.c_int32), ("src_bits", C.c_int32 * 3),
                ("dst", C.c_void_p), ("dst_m", C.c_int32), ("dst_bits", C.c_int32 * 3),
                ("dst_own", C.c_void_p), ("own_pattern", C.c_int32), ("dst_parts", C.c_int32), ("src_parts", C.c_int32),
                ("n_tiles", C.c_int32), ("tile_masks", C.c_void_p)]

    def __init__(self, *args, **kw):
        super().__init__(*args, **kw)
        self.struct_size = C.sizeof(OpsIo)       # (the library refuses a struct of another size)


class OpList(C.Structure):
    """qsim_op_list (include/qsim_hip.h)"""
    _fields_ = [("n_ops", C.c_int32), ("nq", C.c_void_p), ("qubits", C.c_void_p), ("mats", C.c_void_p)]


class ProfileEntry(C.Structure):
    _fields_ = [("kernel", C.c_char * 48), ("launches", C.c_uint64),
                ("total_ms", C.c_double), ("algorithmic_bytes", C.c_double),
                ("hbm_bytes", C.c_double), ("streaming_launches", C.c_uint64)]

_lib = None


def source_hash() -> str:
    """sha256 (16 hex digits) over the kernel sources of libqsim_hip.so: measurement summaries under profiles/
    record it, and bench.py prints a PMC traffic figure only from a summary taken with the same sources."""
    import hashlib
    h = hashlib.sha256()
    csrc = Path(__file__).resolve().parent / "csrc"
    for path in sorted(list(csrc.glob("*.h")) + list(csrc.glob("*.hip")) + list(csrc.glob("*.py")) + [csrc / "Makefile"]):
        h.update(path.name.encode())
        h.update(path.read_bytes())
    return h.hexdigest()[:16]


def load() -> C.CDLL:
    """Load (once) and type the library; raises if it was never built."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise QsimLibraryMissing(
            f"{LIB_PATH} not found: the HIP extension is not built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc). "
            "This engine has no CPU fallback.")
    lib = C.CDLL(str(LIB_PATH))
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so is stale
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def check(rc: int) -> None:
    """Map a C return code to the exception the reference's Python API raises."""
    if rc == QSIM_OK:
        return
    msg = (load().qsim_last_error() or b"").decode("utf-8", "replace")
    if rc == QSIM_ERR_NONLOCAL:
        raise NotImplementedError(msg)  # cpu_scalar.check_local, cpu_scalar.py:13-18
    if rc == QSIM_ERR_INVALID:
        raise ValueError(msg)
    if rc == QSIM_ERR_NOMEM:
        raise MemoryError(msg)
    raise QsimHipError(msg)
