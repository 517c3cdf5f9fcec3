"""numpy interpreter of the fused-pass images that `qsim_plan_ops` emits (test infrastructure).

The device kernel `k_tile` interprets the record stream of a pass image (csrc/tile_kernel.h: group
records, one record per gate = branch-table entry, predicate masks, matrix doubles); this module walks
the SAME stream on a numpy state, following the entry table, so the host planner -- pass building,
register groups, entry and mask encoding, record layout, merged phase runs, ordering -- is checked on
the CPU without a GPU.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from quantum_simulations_amd import _lib
from quantum_simulations_amd.kernel.device import pack_ops

IMAGE_BYTES = 4096
STREAM_OFF = 192                   # byte offset of the first record (csrc/tile_kernel.h)
OPC = dict(DENSE1=1, SWAP1=10, ANTI1=19, PHASE=28, DENSE2=36, REAL1=45, YLIKE1=54, PHASE_NEG=63,
           PHASE_I=71, PHASE_NI=79, DIAGR=87, PRED_OUTER=91, PRED_LANE=92, GROUP=93, GROUP_FIRST=94, END=95,
           HAD1=96, SCALE=105, ASWAP1=106, GROUP_DIRECT=115, END_DIRECT=116, PRED_OUTER_ZERO=117)
DIRECT_IN, DIRECT_OUT = 0x10, 0x20     # TileArgs::order flags
_FAMILIES = ("DENSE1", "SWAP1", "ANTI1", "PHASE", "DENSE2", "REAL1", "YLIKE1", "PHASE_NEG", "PHASE_I", "PHASE_NI", "DIAGR",
             "HAD1", "SCALE", "ASWAP1")
_IMAGE = np.dtype([("amp", "<u8"), ("nrec", "<i4"), ("T", "<i4"), ("h", "u1", (11,)), ("order", "u1"), ("ntiles", "<u4"),
                   ("lay_in", "u1", (12,)), ("lay_out", "u1", (12,)), ("amp_out", "<u8"),
                   # re-layout fused into a pass (planned images carry none: all zero)
                   ("amp_out_own", "<u8"), ("own_mask", "<u8"), ("own_value", "<u8"),
                   ("slab_in", "u1", (40,)), ("slab_out", "u1", (40,)), ("nbits", "u1"), ("perm", "u1"), ("reserved", "u1", (22,)),
                   ("stream", "u1", (IMAGE_BYTES - STREAM_OFF,))])
assert _IMAGE.itemsize == IMAGE_BYTES


def lds_slot(t: int) -> int:
    return t ^ ((t >> 4) & 15)


def records(img):
    """Walk the record stream of a pass image the way the gate engine does (and check the load / store layouts
    of the header against the first / last register group): yields ("group", [s0, s1, s2]) and ("gate", opcode, blk_mask, outer_mask, doubles_at(k), size) tuples."""
    raw = bytes(img.tobytes())
    assert len(raw) == IMAGE_BYTES
    off = STREAM_OFF
    seen_group = False
    count = 0
    groups = []
    T, h = int(img["T"]), [int(x) for x in img["h"]]
    din, dout = bool(int(img["order"]) & DIRECT_IN), bool(int(img["order"]) & DIRECT_OUT)

    def check_layout(lay, direct, s):
        """lay[0..4] thread bits, lay[5..7] register bits of the kernel's global accesses (absolute index bits)"""
        lay = [int(x) for x in lay][:T - 3]
        if not direct:
            assert lay == h[:T - 3], (lay, h)          # through LDS: element tid + 256 j of the tile
            return
        assert T == 11 and s[0] >= 3, "direct layout needs a full tile and a group above the line bits"
        assert lay[5:] == [h[b - 3] for b in s], (lay, s)
        assert lay[:5] == [h[b - 3] for b in range(3, T) if b not in s], (lay, s)

    while True:
        assert off % 16 == 0 and off + 64 <= IMAGE_BYTES, f"record at {off}: the 64-byte fetch leaves the block"
        d = np.frombuffer(raw, dtype="<u4", count=16, offset=off)
        assert d[0] % 4 == 0
        entry, nxt = int(d[0]) // 4, int(d[1])
        count += 1
        if entry in (OPC["END"], OPC["END_DIRECT"]):
            assert nxt == off and seen_group
            assert count == int(img["nrec"]), (count, int(img["nrec"]))
            assert (entry == OPC["END_DIRECT"]) == dout
            check_layout(img["lay_in"], din, groups[0])
            check_layout(img["lay_out"], dout, groups[-1])
            return
        assert off < nxt <= IMAGE_BYTES - 64 and nxt % 16 == 0, (off, nxt)
        if entry in (OPC["GROUP"], OPC["GROUP_FIRST"], OPC["GROUP_DIRECT"]):
            assert (entry != OPC["GROUP"]) == (not seen_group)
            assert (entry == OPC["GROUP_DIRECT"]) == (din and not seen_group)
            seen_group = True
            assert nxt == off + 48
            s = []
            for i in range(3):
                m = int(d[2 + i])
                sh = (m & -m).bit_length() - 1
                assert m == (0xFFFFFFFF << sh) & 0xFFFFFFFF
                s.append(sh)
            for r in range(1, 8):
                t = sum(1 << s[i] for i in range(3) if (r >> i) & 1)
                assert int(d[4 + r]) == lds_slot(t) << 4, "LDS XOR constant"
            groups.append(s)
            yield ("group", s)
        else:
            assert seen_group, "gate before the first group"
            blk, case = int(d[3]) & 0xFFFF, (int(d[3]) >> 16) // 4
            outer = int(d[2]) << 3
            # the second dispatch of a predicated gate may only reach a gate case (the engine's control-flow checks,
            # tests/test_engine_asm_static.py, rely on it)
            assert any(OPC[f] <= case < OPC[f] + (9 if f in ("DENSE1", "SWAP1", "ANTI1", "DENSE2", "REAL1", "YLIKE1", "HAD1", "ASWAP1")
                                                   else 8 if f.startswith("PHASE") else 4 if f == "DIAGR" else 1) for f in _FAMILIES), case
            if entry == OPC["PRED_LANE"]:
                assert blk
            elif entry == OPC["PRED_OUTER_ZERO"]:      # the listed outer bits must all be 0: reported as a NEGATIVE mask
                assert outer and not blk
                outer = -outer
            elif entry == OPC["PRED_OUTER"]:
                assert outer and not blk
            else:
                assert entry == case and not blk and not outer, (entry, case, blk, outer)
            size = nxt - off

            def doubles(k, n, off=off, size=size, raw=raw):
                """n doubles starting at the k-th in-record double; entries beyond the 6th sit at the end"""
                return np.frombuffer(raw, dtype="<f8", count=n, offset=off + 16 + 8 * k)

            def tail(nbytes, off=off, size=size, raw=raw):
                return np.frombuffer(raw, dtype="<f8", count=nbytes // 8, offset=off + size - nbytes)
            yield ("gate", case, blk, outer, doubles, tail, size)
        off = nxt


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


def _apply_1q(psi, idx, t_bit, U, cond_mask, zero_mask=0):
    lo = idx[((idx >> t_bit) & 1 == 0) & ((idx & cond_mask) == cond_mask) & ((idx & zero_mask) == 0)]
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


def _c(v):
    return complex(v[0], v[1])


def fam_of(op: int) -> str:
    base = max(OPC[f] for f in _FAMILIES if OPC[f] <= op)
    return next(k for k in _FAMILIES if OPC[k] == base)


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

    run = 0
    A = None
    s = None
    sunk = []            # OPC_ASWAP1 gates of the current group: they act when the registers are written back

    def write_back():
        for t_bit, cond_mask in sunk:
            _apply_1q(psi, idx, t_bit, np.array([[0, 1], [1, 0]], dtype=complex), cond_mask)
        sunk.clear()

    for rec in records(img):
        if rec[0] == "group":
            write_back()
            s = rec[1]
            assert s[0] < s[1] < s[2] < T, s
            A = [abs_bit(b) for b in s]                                 # absolute index bit of register bit j
            continue
        _, op, blk, outer, dbl, tail, size = rec
        zero_mask = 0
        if outer < 0:                      # OPC_PRED_OUTER_ZERO
            zero_mask, outer = -outer, 0
        assert not (blk & sum(1 << b for b in s)), "lane predicate on a register bit"
        assert not ((outer | zero_mask) & sum(1 << abs_bit(b) for b in range(T))), "outer predicate on a tile bit"
        if zero_mask:                      # only what the planner pairs: plain 1q gates (no register control)
            assert fam_of(op) in ("DENSE1", "REAL1", "ANTI1", "YLIKE1") and op - OPC[fam_of(op)] < 3, op
        cond = abs_mask(blk) | outer
        fam = max(OPC[f] for f in _FAMILIES if OPC[f] <= op)
        var = op - fam
        if fam == OPC["SCALE"]:
            assert var == 0 and not cond and size == 32
            psi *= float(dbl(0, 1)[0])
        elif fam == OPC["ASWAP1"]:
            # deferred X / CNOT: the device swaps LDS addresses, the data moves at the group's write-back -- applied
            # THERE here too, so a gate the planner sank although something later does not commute with it shows
            assert 0 <= var < 9 and size == 16
            if var < 3:
                J, ctrl = var, None
            else:
                J, kk = (var - 3) // 2, (var - 3) % 2
                ctrl = [r for r in range(3) if r != J][kk]
            sunk.append((A[J], cond | (0 if ctrl is None else 1 << A[ctrl])))
        elif fam in (OPC["DENSE1"], OPC["SWAP1"], OPC["ANTI1"], OPC["REAL1"], OPC["YLIKE1"], OPC["HAD1"]):
            assert 0 <= var < 9
            if var < 3:
                J, ctrl = var, None
            else:
                J, kk = (var - 3) // 2, (var - 3) % 2
                ctrl = [r for r in range(3) if r != J][kk]
            if fam == OPC["SWAP1"]:
                U = np.array([[0, 1], [1, 0]], dtype=complex)
                assert size == 16
            elif fam == OPC["YLIKE1"]:
                U = np.array([[0, -1j], [1j, 0]])
                assert size == 16
            elif fam == OPC["HAD1"]:        # unscaled: the pass's SCALE record carries the factors
                U = np.array([[1, 1], [1, -1]], dtype=complex)
                assert size == 16 and var < 3 and not cond
            elif fam == OPC["ANTI1"]:
                m = dbl(0, 4)
                U = np.array([[0, _c(m[0:2])], [_c(m[2:4]), 0]])
                assert size == 48
            elif fam == OPC["REAL1"]:
                m = dbl(0, 4)
                U = np.array([[m[0], m[1]], [m[2], m[3]]], dtype=complex)
                assert size == 48
            else:
                m = dbl(0, 6)
                u11 = tail(16)
                U = np.array([[_c(m[0:2]), _c(m[2:4])], [_c(m[4:6]), _c(u11)]])
                assert size == 80
            _apply_1q(psi, idx, A[J], U, cond | (0 if ctrl is None else 1 << A[ctrl]), zero_mask)
        elif fam in (OPC["PHASE"], OPC["PHASE_NEG"], OPC["PHASE_I"], OPC["PHASE_NI"]):
            assert 0 <= var < 8
            if fam == OPC["PHASE"]:
                f = _c(dbl(0, 2))
                assert size == 32
            else:
                f = {OPC["PHASE_NEG"]: -1, OPC["PHASE_I"]: 1j, OPC["PHASE_NI"]: -1j}[fam]
                assert size == 16
            need = cond | sum(1 << A[r] for r in range(3) if (var >> r) & 1)
            psi[(idx & need) == need] *= f
        elif fam == OPC["DIAGR"]:
            assert 0 <= var < 4
            regs = {0: (0, 1), 1: (0, 2), 2: (1, 2), 3: (0, 1, 2)}[var]
            on = (idx & cond) == cond
            m = dbl(0, 6)
            u = [_c(m[0:2]), _c(m[2:4]), _c(m[4:6])]
            if var < 3:
                assert size == 64 and abs(u[2] - u[0] * u[1]) < 1e-15       # the product the engine uses
                phases = u[:2]
            else:
                assert size == 128
                t = tail(64)
                prods = [_c(t[0:2]), _c(t[2:4]), _c(t[4:6]), _c(t[6:8])]
                want = [u[0] * u[1], u[0] * u[2], u[1] * u[2], u[0] * u[1] * u[2]]
                assert max(abs(a - b) for a, b in zip(prods, want)) < 1e-15
                phases = u
            for e, r in enumerate(regs):
                psi[on & ((idx >> A[r]) & 1 == 1)] *= phases[e]
        elif fam == OPC["DENSE2"]:
            JA, JB = var // 3, var % 3
            assert JA != JB and JA < 3 and size == 272
            m = tail(256)
            _apply_2q(psi, idx, A[JA], A[JB], (m[0::2] + 1j * m[1::2]).reshape(4, 4), cond)
        else:
            raise AssertionError(f"unknown opcode {op}")
        run += 1
    # a direct-out pass stores the last group's registers straight to global memory: there is no write-back through
    # LDS addresses for a sunk swap to act in
    assert not (sunk and int(img["order"]) & DIRECT_OUT), "OPC_ASWAP1 in the last group of a direct-out pass"
    write_back()
    return run


def run(psi: np.ndarray, images) -> int:
    return sum(run_pass(psi, img) for img in images)
