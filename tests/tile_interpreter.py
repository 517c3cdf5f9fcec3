"""numpy interpreter of the fused-pass images that `qsim_plan_ops` emits (test infrastructure).

The device kernel `k_tile` executes a pass image (csrc/tile_kernel.h: group headers + one opcode
byte per gate, predicate masks, matrix pool); this module executes the SAME image on a numpy
state, following the opcode table, so the host planner -- pass building, register groups, opcode
and mask encoding, merged phase runs, ordering -- is checked on the CPU without a GPU.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from quantum_simulations_amd import _lib
from quantum_simulations_amd.kernel.device import pack_ops

IMAGE_BYTES = 4000
MAX_GATES, MAX_MAT = 144, 104      # descriptor slots (143 usable: the device reads one entry ahead), pool entries
OPC = dict(DENSE1=1, SWAP1=10, ANTI1=19, PHASE=28, DENSE2=36, REAL1=45, YLIKE1=54, PHASE_NEG=63,
           PHASE_I=71, PHASE_NI=79, DIAGR=87, GROUP=0xFE)
_GATE = np.dtype([("opcode", "u1"), ("count", "u1"), ("blk_mask", "<u2"), ("mat", "<u2"), ("pad", "<u2"),
                  ("outer_mask", "<u8")])
_IMAGE = np.dtype([("amp", "<u8"), ("ngates", "<i4"), ("T", "<i4"), ("h", "u1", (16,)),
                   ("g", _GATE, (MAX_GATES,)), ("mat", "<c16", (MAX_MAT,))])
assert _IMAGE.itemsize == IMAGE_BYTES


def plan(n_qubits: int, ops) -> np.ndarray:
    """ops [(qubits, U)] -> array of pass images (planned by the C library, no device involved)."""
    nq, qubits, mats = pack_ops(ops)
    lib = _lib.load()
    n_passes = C.c_int32()
    _lib.check(lib.qsim_plan_ops(n_qubits, len(nq), nq.ctypes.data_as(C.c_void_p), qubits.ctypes.data_as(C.c_void_p),
                                 mats.ctypes.data_as(C.c_void_p), None, 0, C.byref(n_passes)))
    out = np.zeros(n_passes.value, dtype=_IMAGE)
    _lib.check(lib.qsim_plan_ops(n_qubits, len(nq), nq.ctypes.data_as(C.c_void_p), qubits.ctypes.data_as(C.c_void_p),
                                 mats.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), out.nbytes,
                                 C.byref(n_passes)))
    return out


def _apply_1q(psi, idx, t_bit, U, cond_mask):
    lo = idx[((idx >> t_bit) & 1 == 0) & ((idx & cond_mask) == cond_mask)]
    hi = lo | (1 << t_bit)
    a, b = psi[lo].copy(), psi[hi].copy()
    psi[lo] = U[0, 0] * a + U[0, 1] * b
    psi[hi] = U[1, 0] * a + U[1, 1] * b


def _apply_2q(psi, idx, qa_bit, qb_bit, U, cond_mask):
    base = idx[((idx >> qa_bit) & 1 == 0) & ((idx >> qb_bit) & 1 == 0) & ((idx & cond_mask) == cond_mask)]
    sel = [base, base | (1 << qb_bit), base | (1 << qa_bit), base | (1 << qa_bit) | (1 << qb_bit)]
    v = np.stack([psi[s] for s in sel])
    r = U @ v
    for s, row in zip(sel, r):
        psi[s] = row


def run_pass(psi: np.ndarray, img) -> int:
    """Execute one pass image on `psi` in place; returns the number of gate descriptors run."""
    T = int(img["T"])
    n = int(np.log2(psi.size))
    idx = np.arange(psi.size, dtype=np.int64)
    h = [int(x) for x in img["h"][:T - 3]]
    assert sorted(set(h)) == h and all(3 <= b < n for b in h), h       # ascending, distinct, above the low bits

    def abs_bit(tile_bit: int) -> int:
        return tile_bit if tile_bit < 3 else h[tile_bit - 3]

    def abs_mask(tile_mask: int) -> int:
        return sum(1 << abs_bit(b) for b in range(T) if (tile_mask >> b) & 1)

    g, mat, ngates = img["g"], img["mat"], int(img["ngates"])
    i = 0
    run = 0
    while i < ngates:
        hd = g[i]
        assert hd["opcode"] == OPC["GROUP"], (i, hd["opcode"])
        s = [(int(hd["blk_mask"]) >> (4 * j)) & 15 for j in range(3)]
        assert s[0] < s[1] < s[2] < T, s
        A = [abs_bit(b) for b in s]                                     # absolute index bit of register bit j
        for d in g[i + 1:i + 1 + int(hd["count"])]:
            op, m = int(d["opcode"]), int(d["mat"])
            assert not (int(d["blk_mask"]) & sum(1 << b for b in s)), "lane predicate on a register bit"
            cond = abs_mask(int(d["blk_mask"])) | int(d["outer_mask"])
            fam = max(v for v in OPC.values() if v <= op and v != OPC["GROUP"])
            var = op - fam
            if fam in (OPC["DENSE1"], OPC["SWAP1"], OPC["ANTI1"], OPC["REAL1"], OPC["YLIKE1"]):
                if var < 3:
                    J, ctrl = var, None
                else:
                    J, kk = (var - 3) // 2, (var - 3) % 2
                    ctrl = [r for r in range(3) if r != J][kk]
                if fam == OPC["SWAP1"]:
                    U = np.array([[0, 1], [1, 0]], dtype=complex)
                elif fam == OPC["YLIKE1"]:
                    U = np.array([[0, -1j], [1j, 0]])
                elif fam == OPC["ANTI1"]:
                    U = np.array([[0, mat[m + 1]], [mat[m + 2], 0]])
                elif fam == OPC["REAL1"]:       # packed: entry 0 = (r00, r01), entry 1 = (r10, r11)
                    U = np.array([[mat[m].real, mat[m].imag], [mat[m + 1].real, mat[m + 1].imag]], dtype=complex)
                else:
                    U = mat[m:m + 4].reshape(2, 2)
                _apply_1q(psi, idx, A[J], U, cond | (0 if ctrl is None else 1 << A[ctrl]))
            elif fam in (OPC["PHASE"], OPC["PHASE_NEG"], OPC["PHASE_I"], OPC["PHASE_NI"]):
                f = {OPC["PHASE"]: mat[m], OPC["PHASE_NEG"]: -1, OPC["PHASE_I"]: 1j, OPC["PHASE_NI"]: -1j}[fam]
                need = cond | sum(1 << A[r] for r in range(3) if (var >> r) & 1)
                psi[(idx & need) == need] *= f
            elif fam == OPC["DIAGR"]:
                regs = {0: (0, 1), 1: (0, 2), 2: (1, 2), 3: (0, 1, 2)}[var]
                on = (idx & cond) == cond
                for e, r in enumerate(regs):
                    psi[on & ((idx >> A[r]) & 1 == 1)] *= mat[m + e]
            elif fam == OPC["DENSE2"]:
                JA, JB = var // 3, var % 3
                assert JA != JB
                _apply_2q(psi, idx, A[JA], A[JB], mat[m:m + 16].reshape(4, 4), cond)
            else:
                raise AssertionError(f"unknown opcode {op}")
            run += 1
        i += 1 + int(hd["count"])
    return run


def run(psi: np.ndarray, images) -> int:
    return sum(run_pass(psi, img) for img in images)
