#!/usr/bin/env python3
"""Generator of the gate engine of k_tile: the compute phase of a fused tile pass, written as
gfx950 (CDNA4) assembly and emitted as ONE inline-asm block (tile_engine_gen.h).

Why assembly: the C++ form of the gate loop (r01) spent ~27 scalar instructions per gate
descriptor on decode + a 7-level compare/branch tree (hipcc has no jump tables on AMDGPU and
`asm goto` is not lowered), and ~45 vector/scalar instructions per register-group change on LDS
address arithmetic.  The pass was bound by total instruction issue (rocprofv3 SQ counters:
1247 SALU + 1137 VALU + 98 SMEM per wave and pass, time ~ 4 cycles x instructions x waves/SIMD).

The engine is a *banked, direct-threaded interpreter* of a record stream in the kernel-argument
block (tile_kernel.h, `TileArgs::stream`):

  * a record = 16-byte header + matrix doubles; every record is fetched with ONE
    `s_load_dwordx16` (header + up to 6 doubles) ONE RECORD AHEAD into the *other* SGPR bank
    (bank A = s[36:51], bank B = s[52:67]); records alternate banks, so every case exists twice
    (one body per bank) and nothing is ever copied;
  * dispatch = `s_add_u32 / s_addc_u32 / s_setpc_b64` into a table of `s_branch` (header dword 0 is
    4 x entry, written by the host);
  * predicates (index bits outside the tile: `outer`; tile bits outside the register group:
    `lane`) have their own entries that test, narrow EXEC by hand and dispatch a second time, so
    plain gates pay nothing for them;
  * a register-group change is a record too: the host supplies the three insert-zero masks and the
    seven XOR constants of the swizzled LDS addresses (the swizzle t ^ ((t >> 4) & 15) is linear
    over GF(2), so slot(tb | b) = slot(tb) ^ slot(b) with slot(b) wave-uniform).

Fixed registers (clobbers of the asm statement, except x0..x7: in/out operands pinned to v[4:35], because the
first / last register group of a pass may be the layout the kernel loads / stores in -- GROUP_DIRECT starts
from the operands' values without reading LDS, END_DIRECT leaves the result in them without writing it):
  v[4:35]   x0..x7 (complex128 each: .x = v[4+4j:5+4j], .y = v[6+4j:7+4j])
  v[36:43]  LDS byte addresses of x0..x7      v44 tb (thread's tile index with the group bits 0)
  v45 scratch   v[46:61] eight f64 temporaries   (the thread id is read from the operand's register)
  s[16:17] scratch pair  s18 scratch  s19 base >> 3 (outer predicate)
  s[20:21] / s[22:23] branch-table base of bank A / B   s[24:25] jump target
  s[26:27] kernel-argument pointer   s[28:29] mask of live lanes (tiles smaller than 8 x blockDim)
  s[36:51] bank A   s[52:67] bank B   s[68:83] overflow matrix entries (dense 2x2 / 4x4, 3-phase runs)

Run:  python3 gen_tile_engine.py > tile_engine_gen.h      (the Makefile does; the result is committed)
"""
from __future__ import annotations

import os
import sys

# Generation-time variants (A/B builds): QS_GEN_TAILS=1 ends every gate body with the fetch + dispatch
# of the next record instead of a branch to the shared copy of that sequence.
TAILS = os.environ.get("QS_GEN_TAILS", "0") == "1"
# PROBE ONLY (wrong results): QS_GEN_NOBARRIER=1 drops the barrier of register-group changes, to
# measure what the barriers cost.
NOBARRIER = os.environ.get("QS_GEN_NOBARRIER", "0") == "1"
# PROBE ONLY: extra instructions per record in the fetch + dispatch sequence, e.g. QS_GEN_PAD="s_nop 0*4"
PAD = os.environ.get("QS_GEN_PAD", "")
PAD_AFTER = os.environ.get("QS_GEN_PAD_AFTER", "")      # ... the same after the wait for the record
# A/B: QS_GEN_SWAP=mov64 exchanges amplitude registers with v_mov_b64 through the f64 temporaries (6 full-rate moves
# per pair of amplitudes) instead of 4 v_swap_b32 (half rate)
SWAP_MOV64 = os.environ.get("QS_GEN_SWAP", "") == "mov64"


# ---- entry numbers (header dword 0 = 4 * entry); the gate families keep the r01 opcode numbers ----
OPC = dict(NOP=0, DENSE1=1, SWAP1=10, ANTI1=19, PHASE=28, DENSE2=36, REAL1=45, YLIKE1=54,
           PHASE_NEG=63, PHASE_I=71, PHASE_NI=79, DIAGR=87,
           PRED_OUTER=91, PRED_LANE=92, GROUP=93, GROUP_FIRST=94, END=95, HAD1=96, SCALE=105, ASWAP1=106,
           GROUP_DIRECT=115, END_DIRECT=116, PRED_OUTER_ZERO=117)
NENT = 118

BANK = {"A": 36, "B": 52}                # s32 (the ABI stack pointer) is reserved: banks start at s36
E = 68                                   # overflow bank s[68:83]


def X(j):            # (x pair, y pair, quad) of amplitude register j
    b = 4 + 4 * j
    return f"v[{b}:{b + 1}]", f"v[{b + 2}:{b + 3}]", f"v[{b}:{b + 3}]"


def XD(j, c):        # dword c (0..3) of amplitude register j
    return f"v{4 + 4 * j + c}"


def T(k):
    return f"v[{46 + 2 * k}:{47 + 2 * k}]"


def A(j):
    return f"v{36 + j}"


TB, VT, TID = "v44", "v45", "%[tid]"     # (the thread id stays in the compiler's register)


def M(bank, k):      # k-th in-record double of the bank's record
    b = BANK[bank] + 4 + 2 * k
    return f"s[{b}:{b + 1}]"


def EM(k):           # k-th double of the overflow bank
    return f"s[{E + 2 * k}:{E + 2 * k + 1}]"


def HD(bank, d):     # header dword d (also group-record dwords 4..11)
    return f"s{BANK[bank] + d}"


PAIRS = {0: [(0, 1), (2, 3), (4, 5), (6, 7)], 1: [(0, 2), (1, 3), (4, 6), (5, 7)], 2: [(0, 4), (1, 5), (2, 6), (3, 7)],
         3: [(2, 3), (6, 7)], 4: [(4, 5), (6, 7)], 5: [(1, 3), (5, 7)], 6: [(4, 6), (5, 7)], 7: [(1, 5), (3, 7)],
         8: [(2, 6), (3, 7)]}
PHASE_REGS = {0: [0, 1, 2, 3, 4, 5, 6, 7], 1: [1, 3, 5, 7], 2: [2, 3, 6, 7], 3: [3, 7], 4: [4, 5, 6, 7], 5: [5, 7],
              6: [6, 7], 7: [7]}


class Asm:
    def __init__(self):
        self.lines = []

    def __call__(self, text):
        self.lines.append(text)

    def label(self, name):
        self.lines.append(f"{lab(name)}:")


def lab(name):
    return f".Lqs_{name}_%="


def other(bank):
    return "B" if bank == "A" else "A"


# ---- straight-line gate bodies -------------------------------------------------------------------
def chunks(seq, n):
    return [seq[i:i + n] for i in range(0, len(seq), n)]


def body_real(a, bank, pairs):
    """x_a' = r00 x_a + r01 x_b, x_b' = r10 x_a + r11 x_b (real entries r00,r01,r10,r11 = doubles 0..3)"""
    r00, r01, r10, r11 = (M(bank, k) for k in range(4))
    for grp in chunks(pairs, 2):
        for i, (pa, pb) in enumerate(grp):
            ax, ay, _ = X(pa)
            a(f"v_mul_f64 {T(4 * i)}, {r00}, {ax}")
            a(f"v_mul_f64 {T(4 * i + 1)}, {r00}, {ay}")
            a(f"v_mul_f64 {T(4 * i + 2)}, {r10}, {ax}")
            a(f"v_mul_f64 {T(4 * i + 3)}, {r10}, {ay}")
        for i, (pa, pb) in enumerate(grp):
            ax, ay, _ = X(pa)
            bx, by, _ = X(pb)
            a(f"v_fma_f64 {ax}, {r01}, {bx}, {T(4 * i)}")
            a(f"v_fma_f64 {ay}, {r01}, {by}, {T(4 * i + 1)}")
            a(f"v_fma_f64 {bx}, {r11}, {bx}, {T(4 * i + 2)}")
            a(f"v_fma_f64 {by}, {r11}, {by}, {T(4 * i + 3)}")


def body_had(a, bank, pairs):
    """unscaled Hadamard butterfly x_a' = x_a + x_b, x_b' = x_a - x_b; the factor c of c [[1,1],[1,-1]] is
    collected by the host over the pass and applied once (OPC_SCALE): b' = a - b, then a' = 2a - b'"""
    for pa, pb in pairs:
        ax, ay, _ = X(pa)
        bx, by, _ = X(pb)
        a(f"v_add_f64 {bx}, {ax}, -{bx}")
        a(f"v_add_f64 {by}, {ay}, -{by}")
    for pa, pb in pairs:
        ax, ay, _ = X(pa)
        bx, by, _ = X(pb)
        a(f"v_fma_f64 {ax}, 2.0, {ax}, -{bx}")
        a(f"v_fma_f64 {ay}, 2.0, {ay}, -{by}")


def body_scale(a, bank):
    """every amplitude of the tile times the real double 0 of the record"""
    for r in range(8):
        xx, xy, _ = X(r)
        a(f"v_mul_f64 {xx}, {M(bank, 0)}, {xx}")
        a(f"v_mul_f64 {xy}, {M(bank, 0)}, {xy}")


def body_dense1(a, bank, pairs):
    """general complex 2x2: u00 = doubles (0,1), u01 = (2,3), u10 = (4,5) of the record, u11 = the 16
    bytes in front of the next record (loaded here into the overflow bank)"""
    a(f"s_sub_u32 s18, {HD(bank, 1)}, 16")
    a(f"s_load_dwordx4 s[{E}:{E + 3}], s[26:27], s18")
    u00x, u00y, u01x, u01y, u10x, u10y = (M(bank, k) for k in range(6))
    u11x, u11y = EM(0), EM(1)
    first = True
    for grp in chunks(pairs, 2):
        steps = []
        for i, (pa, pb) in enumerate(grp):
            ax, ay, _ = X(pa)
            bx, by, _ = X(pb)
            t0, t1, t2, t3 = (T(4 * i + c) for c in range(4))
            steps.append([
                f"v_mul_f64 {t0}, {u00x}, {ax}", f"v_mul_f64 {t1}, {u00x}, {ay}",
                f"v_mul_f64 {t2}, {u10x}, {ax}", f"v_mul_f64 {t3}, {u10x}, {ay}",
                f"v_fma_f64 {t0}, -{u00y}, {ay}, {t0}", f"v_fma_f64 {t1}, {u00y}, {ax}, {t1}",
                f"v_fma_f64 {t2}, -{u10y}, {ay}, {t2}", f"v_fma_f64 {t3}, {u10y}, {ax}, {t3}",
                f"v_fma_f64 {t0}, {u01x}, {bx}, {t0}", f"v_fma_f64 {t1}, {u01x}, {by}, {t1}",
                f"v_fma_f64 {ax}, -{u01y}, {by}, {t0}", f"v_fma_f64 {ay}, {u01y}, {bx}, {t1}",
                "WAIT",
                f"v_fma_f64 {t2}, {u11x}, {bx}, {t2}", f"v_fma_f64 {t3}, {u11x}, {by}, {t3}",
                f"v_fma_f64 {t3}, {u11y}, {bx}, {t3}", f"v_fma_f64 {bx}, -{u11y}, {by}, {t2}",
                f"v_mov_b64 {by}, {t3}"])
        for row in zip(*steps):
            if row[0] == "WAIT":
                if first:
                    a("s_waitcnt lgkmcnt(0)")
                    first = False
                continue
            for ins in row:
                a(ins)


def body_anti(a, bank, pairs):
    """x_a' = u01 x_b, x_b' = u10 x_a; u01 = doubles (0,1), u10 = (2,3)"""
    u01x, u01y, u10x, u10y = (M(bank, k) for k in range(4))
    for grp in chunks(pairs, 3):
        for i, (pa, pb) in enumerate(grp):
            bx, by, _ = X(pb)
            a(f"v_mul_f64 {T(2 * i)}, {u01x}, {bx}")
            a(f"v_mul_f64 {T(2 * i + 1)}, {u01x}, {by}")
        for i, (pa, pb) in enumerate(grp):
            bx, by, _ = X(pb)
            a(f"v_fma_f64 {T(2 * i)}, -{u01y}, {by}, {T(2 * i)}")
            a(f"v_fma_f64 {T(2 * i + 1)}, {u01y}, {bx}, {T(2 * i + 1)}")
        for i, (pa, pb) in enumerate(grp):
            ax, ay, _ = X(pa)
            bx, by, _ = X(pb)
            a(f"v_mul_f64 {bx}, {u10x}, {ax}")
            a(f"v_mul_f64 {by}, {u10x}, {ay}")
        for i, (pa, pb) in enumerate(grp):
            ax, ay, _ = X(pa)
            bx, by, _ = X(pb)
            a(f"v_fma_f64 {bx}, -{u10y}, {ay}, {bx}")
            a(f"v_fma_f64 {by}, {u10y}, {ax}, {by}")
        for i, (pa, pb) in enumerate(grp):
            ax, ay, _ = X(pa)
            a(f"v_mov_b64 {ax}, {T(2 * i)}")
            a(f"v_mov_b64 {ay}, {T(2 * i + 1)}")


def body_swap(a, bank, pairs):
    if SWAP_MOV64:
        for i, (pa, pb) in enumerate(pairs):
            ax, ay, _ = X(pa)
            a(f"v_mov_b64 {T(2 * i)}, {ax}")
            a(f"v_mov_b64 {T(2 * i + 1)}, {ay}")
        for pa, pb in pairs:
            ax, ay, _ = X(pa)
            bx, by, _ = X(pb)
            a(f"v_mov_b64 {ax}, {bx}")
            a(f"v_mov_b64 {ay}, {by}")
        for i, (pa, pb) in enumerate(pairs):
            bx, by, _ = X(pb)
            a(f"v_mov_b64 {bx}, {T(2 * i)}")
            a(f"v_mov_b64 {by}, {T(2 * i + 1)}")
        return
    for pa, pb in pairs:
        for c in range(4):
            a(f"v_swap_b32 {XD(pa, c)}, {XD(pb, c)}")


def body_aswap(a, bank, pairs):
    """X / CNOT deferred to the write-back: the LDS ADDRESSES of the pair trade places (one v_swap_b32 per pair
    instead of four on the data).  The host uses it only when nothing later in the group touches the target
    (or a register control): the registers then reach LDS at the swapped positions."""
    for pa, pb in pairs:
        a(f"v_swap_b32 {A(pa)}, {A(pb)}")


def body_ylike(a, bank, pairs):
    """[[0,-i],[i,0]]: x_a' = -i x_b = (b.y, -b.x), x_b' = i x_a = (-a.y, a.x)"""
    for pa, pb in pairs:
        ax, ay, _ = X(pa)
        bx, by, _ = X(pb)
        # a.x <-> b.y, then a.y <-> b.x, then negate a.y and b.x
        a(f"v_swap_b32 {XD(pa, 0)}, {XD(pb, 2)}")
        a(f"v_swap_b32 {XD(pa, 1)}, {XD(pb, 3)}")
        a(f"v_swap_b32 {XD(pa, 2)}, {XD(pb, 0)}")
        a(f"v_swap_b32 {XD(pa, 3)}, {XD(pb, 1)}")
        a(f"v_mul_f64 {ay}, -1.0, {ay}")
        a(f"v_mul_f64 {bx}, -1.0, {bx}")


def phase_ops(a, regs_and_u):
    """x *= u for a list of (register, (ux, uy)); u operands are SGPR or VGPR pairs"""
    for grp in chunks(regs_and_u, 4):
        for i, (r, (ux, uy)) in enumerate(grp):
            xx, xy, _ = X(r)
            a(f"v_mul_f64 {T(2 * i)}, {ux}, {xx}")
            a(f"v_mul_f64 {T(2 * i + 1)}, {uy}, {xx}")
        for i, (r, (ux, uy)) in enumerate(grp):
            xx, xy, _ = X(r)
            a(f"v_fma_f64 {xx}, -{uy}, {xy}, {T(2 * i)}")
            a(f"v_fma_f64 {xy}, {ux}, {xy}, {T(2 * i + 1)}")


def body_phase(a, bank, regs):
    phase_ops(a, [(r, (M(bank, 0), M(bank, 1))) for r in regs])


def body_phase_neg(a, bank, regs):
    for r in regs:
        xx, xy, _ = X(r)
        a(f"v_mul_f64 {xx}, -1.0, {xx}")
        a(f"v_mul_f64 {xy}, -1.0, {xy}")


def body_phase_i(a, bank, regs):       # x = i x = (-x.y, x.x)
    for r in regs:
        xx, xy, _ = X(r)
        a(f"v_swap_b32 {XD(r, 0)}, {XD(r, 2)}")
        a(f"v_swap_b32 {XD(r, 1)}, {XD(r, 3)}")
        a(f"v_mul_f64 {xx}, -1.0, {xx}")


def body_phase_ni(a, bank, regs):      # x = -i x = (x.y, -x.x)
    for r in regs:
        xx, xy, _ = X(r)
        a(f"v_swap_b32 {XD(r, 0)}, {XD(r, 2)}")
        a(f"v_swap_b32 {XD(r, 1)}, {XD(r, 3)}")
        a(f"v_mul_f64 {xy}, -1.0, {xy}")


def body_diagr(a, bank, var):
    """several phase gates that share their predicate, one per register bit (merged by the host)"""
    if var < 3:        # two bits (p, q): record doubles = u_p, u_q, u_p * u_q
        p, q = {0: (0, 1), 1: (0, 2), 2: (1, 2)}[var]
        up, uq, w = (M(bank, 0), M(bank, 1)), (M(bank, 2), M(bank, 3)), (M(bank, 4), M(bank, 5))
        ops = []
        for r in range(8):
            bp, bq = (r >> p) & 1, (r >> q) & 1
            if bp and bq:
                ops.append((r, w))
            elif bp:
                ops.append((r, up))
            elif bq:
                ops.append((r, uq))
        phase_ops(a, ops)
        return
    # three bits: a, b, c in the record; ab, ac, bc, abc in the 64 bytes in front of the next record
    a(f"s_sub_u32 s18, {HD(bank, 1)}, 64")
    a(f"s_load_dwordx16 s[{E}:{E + 15}], s[26:27], s18")
    a("s_waitcnt lgkmcnt(0)")
    ua, ub, uc = (M(bank, 0), M(bank, 1)), (M(bank, 2), M(bank, 3)), (M(bank, 4), M(bank, 5))
    uab, uac, ubc, uabc = ((EM(2 * k), EM(2 * k + 1)) for k in range(4))
    phase_ops(a, [(1, ua), (2, ub), (4, uc), (3, uab), (5, uac), (6, ubc), (7, uabc)])


def body_dense2(a, bank, ja, jb):
    """general 4x4 on register bits (ja = MSB of the 4-index, jb = LSB); the 16 complex entries sit in
    the 256 bytes in front of the next record and are fetched one row (4 entries) at a time"""
    rest = [r for r in range(3) if r not in (ja, jb)][0]
    a(f"s_sub_u32 s18, {HD(bank, 1)}, 256")
    for fixed in (0, 1):
        base = fixed << rest
        quad = [base, base | (1 << jb), base | (1 << ja), base | (1 << ja) | (1 << jb)]
        for r in range(4):                     # row r -> temporaries 2r, 2r+1
            a(f"s_load_dwordx16 s[{E}:{E + 15}], s[26:27], s18 offset:{64 * r}")
            a("s_waitcnt lgkmcnt(0)")
            dx, dy = T(2 * r), T(2 * r + 1)
            for j, q in enumerate(quad):
                mx, my = EM(2 * j), EM(2 * j + 1)
                xx, xy, _ = X(q)
                if j == 0:
                    a(f"v_mul_f64 {dx}, {mx}, {xx}")
                    a(f"v_mul_f64 {dy}, {mx}, {xy}")
                else:
                    a(f"v_fma_f64 {dx}, {mx}, {xx}, {dx}")
                    a(f"v_fma_f64 {dy}, {mx}, {xy}, {dy}")
                a(f"v_fma_f64 {dx}, -{my}, {xy}, {dx}")
                a(f"v_fma_f64 {dy}, {my}, {xx}, {dy}")
        for r, q in enumerate(quad):
            xx, xy, _ = X(q)
            a(f"v_mov_b64 {xx}, {T(2 * r)}")
            a(f"v_mov_b64 {xy}, {T(2 * r + 1)}")


def gate_cases():
    """entry -> function(asm, bank) emitting the straight-line body"""
    cases = {}
    for fam, body in (("DENSE1", body_dense1), ("SWAP1", body_swap), ("ANTI1", body_anti), ("REAL1", body_real),
                      ("YLIKE1", body_ylike), ("HAD1", body_had), ("ASWAP1", body_aswap)):
        for var in range(9):
            cases[OPC[fam] + var] = (f"{fam.lower()}_{var}", lambda a, bank, body=body, var=var: body(a, bank, PAIRS[var]))
    for fam, body in (("PHASE", body_phase), ("PHASE_NEG", body_phase_neg), ("PHASE_I", body_phase_i),
                      ("PHASE_NI", body_phase_ni)):
        for var in range(8):
            cases[OPC[fam] + var] = (f"{fam.lower()}_{var}", lambda a, bank, body=body, var=var: body(a, bank, PHASE_REGS[var]))
    cases[OPC["SCALE"]] = ("scale", body_scale)
    for var in range(4):
        cases[OPC["DIAGR"] + var] = (f"diagr_{var}", lambda a, bank, var=var: body_diagr(a, bank, var))
    for ja in range(3):
        for jb in range(3):
            if ja != jb:
                cases[OPC["DENSE2"] + 3 * ja + jb] = (f"dense2_{ja}{jb}", lambda a, bank, ja=ja, jb=jb: body_dense2(a, bank, ja, jb))
    return cases


# ---- the engine ---------------------------------------------------------------------------------
def dispatch(a, bank, off_reg):
    jb = 20 if bank == "A" else 22
    a(f"s_add_u32 s24, s{jb}, {off_reg}")
    a(f"s_addc_u32 s25, s{jb + 1}, 0")
    a("s_setpc_b64 s[24:25]")


def top_sequence(a, bank):
    """Record held in `bank` has arrived (after the wait): fetch the next one into the other bank, dispatch."""
    nb = BANK[other(bank)]
    a("s_mov_b64 exec, -1")
    if PAD:
        ins, _, cnt = PAD.partition("*")
        for _ in range(int(cnt or 1)):
            a(ins)
    a("s_waitcnt lgkmcnt(0)")
    if PAD_AFTER:
        ins, _, cnt = PAD_AFTER.partition("*")
        for _ in range(int(cnt or 1)):
            a(ins)
    a(f"s_load_dwordx16 s[{nb}:{nb + 15}], s[26:27], {HD(bank, 1)}")
    dispatch(a, bank, HD(bank, 0))


def next_record(a, bank):
    """End of a gate body that ran from `bank`: go on with the record in the other bank."""
    if TAILS:
        top_sequence(a, other(bank))
    else:
        a(f"s_branch {lab('top_' + other(bank))}")


def engine(partial: bool) -> list[str]:
    """partial: tiles with fewer than 8 x blockDim amplitudes -- only lanes tid < NBLK own a register
    block (operand %[nblk]); they alone touch LDS."""
    a = Asm()
    cases = gate_cases()
    a("s_mov_b64 s[26:27], %[karg]")
    a("s_mov_b32 s19, %[baseh]")
    if partial:
        a(f"v_cmp_gt_u32_e64 s[28:29], %[nblk], {TID}")
    a("s_getpc_b64 s[20:21]")
    a(f"s_branch {lab('start')}")
    for bank in "AB":                      # branch tables: entry e of bank X at table_X + 4 e
        for e in range(NENT):
            if e in cases:
                a(f"s_branch {lab(cases[e][0] + '_' + bank)}")
            elif e == OPC["NOP"]:
                a(f"s_branch {lab('top_' + other(bank))}")
            elif e == OPC["PRED_OUTER"]:
                a(f"s_branch {lab('pred_outer_' + bank)}")
            elif e == OPC["PRED_LANE"]:
                a(f"s_branch {lab('pred_lane_' + bank)}")
            elif e == OPC["PRED_OUTER_ZERO"]:
                a(f"s_branch {lab('pred_outer_zero_' + bank)}")
            elif e == OPC["GROUP"]:
                a(f"s_branch {lab('group_' + bank)}")
            elif e == OPC["GROUP_FIRST"]:
                a(f"s_branch {lab('group_first_' + bank)}")
            elif e == OPC["GROUP_DIRECT"]:
                a(f"s_branch {lab('group_direct_' + bank)}")
            elif e == OPC["END_DIRECT"]:
                a(f"s_branch {lab('end_direct')}")
            else:                          # END and unused entries
                a(f"s_branch {lab('end')}")
    a.label("start")
    a("s_add_u32 s20, s20, 4")             # s_getpc gave the address of the s_branch in front of table A
    a("s_addc_u32 s21, s21, 0")
    a(f"s_add_u32 s22, s20, {4 * NENT}")
    a("s_addc_u32 s23, s21, 0")
    a(f"s_load_dwordx16 s[{BANK['A']}:{BANK['A'] + 15}], s[26:27], %[first]")
    for bank in "AB":
        nb = BANK[other(bank)]
        # ---- top of a record held in `bank`: fetch the next one into the other bank, dispatch ----
        a.label("top_" + bank)
        top_sequence(a, bank)
        # ---- predicates ----
        a.label("pred_lane_" + bank)
        a(f"s_and_b32 s18, {HD(bank, 2)}, s19")
        a(f"s_cmp_lg_u32 s18, {HD(bank, 2)}")
        a(f"s_cbranch_scc1 {lab('top_' + other(bank))}")
        a(f"s_and_b32 s18, {HD(bank, 3)}, 0xffff")
        a(f"v_and_b32 {VT}, s18, {TB}")
        a(f"v_cmp_eq_u32_e64 s[16:17], s18, {VT}")
        a("s_and_b64 exec, exec, s[16:17]")
        a(f"s_cbranch_scc0 {lab('top_' + other(bank))}")
        a(f"s_lshr_b32 s18, {HD(bank, 3)}, 16")
        dispatch(a, bank, "s18")
        a.label("pred_outer_" + bank)
        a(f"s_and_b32 s18, {HD(bank, 2)}, s19")
        a(f"s_cmp_lg_u32 s18, {HD(bank, 2)}")
        a(f"s_cbranch_scc1 {lab('top_' + other(bank))}")
        a(f"s_lshr_b32 s18, {HD(bank, 3)}, 16")
        dispatch(a, bank, "s18")
        # ... and its complement: the listed index bits outside the tile must all be 0 (the control = 0 half of a
        # controlled gate that was merged with the 1q gate next to it on its target, tile_planner.h)
        a.label("pred_outer_zero_" + bank)
        a(f"s_and_b32 s18, {HD(bank, 2)}, s19")
        a("s_cmp_lg_u32 s18, 0")
        a(f"s_cbranch_scc1 {lab('top_' + other(bank))}")
        a(f"s_lshr_b32 s18, {HD(bank, 3)}, 16")
        dispatch(a, bank, "s18")
        # ---- register-group change: write the registers back, barrier, pull the next group ----
        a.label("group_" + bank)
        if partial:
            a("s_mov_b64 exec, s[28:29]")
        for j in range(8):
            a(f"ds_write_b128 {A(j)}, {X(j)[2]}")
        a("s_waitcnt lgkmcnt(0)")
        if partial:
            a("s_mov_b64 exec, -1")
        a("s_nop 0" if NOBARRIER else "s_barrier")
        def addresses():
            # record dwords: 2..4 = insert-zero masks (~0 << s_i, ascending), 5..11 = XOR constants of x1..x7
            a(f"v_and_b32 {VT}, {HD(bank, 2)}, {TID}")
            a(f"v_add_u32 {TB}, {TID}, {VT}")
            a(f"v_and_b32 {VT}, {HD(bank, 3)}, {TB}")
            a(f"v_add_u32 {TB}, {TB}, {VT}")
            a(f"v_and_b32 {VT}, {HD(bank, 4)}, {TB}")
            a(f"v_add_u32 {TB}, {TB}, {VT}")
            a(f"v_bfe_u32 {VT}, {TB}, 4, 4")
            a(f"v_xor_b32 {A(0)}, {VT}, {TB}")
            a(f"v_lshlrev_b32 {A(0)}, 4, {A(0)}")
            for j in range(1, 8):
                a(f"v_xor_b32 {A(j)}, {HD(bank, 4 + j)}, {A(0)}")
        a.label("group_first_" + bank)
        addresses()
        if partial:
            a("s_mov_b64 exec, s[28:29]")
        for j in range(8):
            a(f"ds_read_b128 {X(j)[2]}, {A(j)}")
        a(f"s_branch {lab('top_' + other(bank))}")
        # ---- first group = the layout the kernel loaded in: x0..x7 hold it already, only the addresses of
        # the write-back are needed ----
        a.label("group_direct_" + bank)
        addresses()
        a(f"s_branch {lab('top_' + other(bank))}")
        # ---- gate bodies ----
        for e in sorted(cases):
            name, fn = cases[e]
            a.label(name + "_" + bank)
            fn(a, bank)
            next_record(a, bank)
    # ---- end of the stream: last write-back; the tile is complete in LDS after the barrier ----
    a.label("end")
    if partial:
        a("s_mov_b64 exec, s[28:29]")
    else:
        a("s_mov_b64 exec, -1")
    for j in range(8):
        a(f"ds_write_b128 {A(j)}, {X(j)[2]}")
    a("s_mov_b64 exec, -1")
    a("s_waitcnt lgkmcnt(0)")
    a("s_barrier")
    # ---- END_DIRECT: the last group is the layout the kernel stores in; the result stays in x0..x7 ----
    a.label("end_direct")
    a("s_mov_b64 exec, -1")
    # the look-ahead fetch of the record "after" END is still in flight: it must land before the statement
    # ends, or it overwrites whatever the compiler puts into the (clobbered, hence free) bank registers next
    a("s_waitcnt lgkmcnt(0)")
    return a.lines


def clobbers() -> str:
    regs = [f"v{i}" for i in range(36, 62)] + [f"s{i}" for i in range(16, 30)] + [f"s{i}" for i in range(36, 84)] + ["vcc", "scc", "memory"]
    return ", ".join(f'"{r}"' for r in regs)


def c_string(lines) -> str:
    return "\n".join(f'  "{ln}\\n"' for ln in lines)


def main():
    probes = sorted(k for k in os.environ if k.startswith("QS_GEN_") and k != "QS_GEN_ALLOW_PROBES")
    if probes and os.environ.get("QS_GEN_ALLOW_PROBES") != "1":
        sys.exit(f"gen_tile_engine.py: probe options {probes} are set; they change (some of them break) the engine. "
                 "Set QS_GEN_ALLOW_PROBES=1 for an A/B build, never for the committed header.")
    out = sys.stdout
    out.write("// tile_engine_gen.h -- GENERATED by gen_tile_engine.py (do not edit): the gate engine of k_tile\n"
              "// as one gfx950 inline-asm block.  See the generator for the design and the register map.\n"
              "#pragma once\n")
    for name, value in OPC.items():
        out.write(f"#define QS_ENT_{name} {value}\n")
    out.write(f"#define QS_ENGINE_NENT {NENT}\n")
    out.write(f"#define QS_ENGINE_CLOBBERS {clobbers()}\n")
    out.write("// operands: %[tid] v, %[karg] s (64-bit), %[baseh] s, %[first] i (byte offset of the first record)\n")
    out.write("#define QS_ENGINE_ASM_FULL \\\n" + c_string(engine(False)).replace("\n", " \\\n") + "\n")
    out.write("// + %[nblk] s: register blocks per tile (tiles smaller than 8 x blockDim)\n")
    out.write("#define QS_ENGINE_ASM_PARTIAL \\\n" + c_string(engine(True)).replace("\n", " \\\n") + "\n")


if __name__ == "__main__":
    main()
