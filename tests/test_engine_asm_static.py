"""Static checks of the generated gate engine (csrc/gen_tile_engine.py -> tile_engine_gen.h): things the
assembler does not reject but the hardware or the surrounding C++ would get wrong silently.  No GPU needed."""
import re
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "quantum_simulations_amd" / "csrc"))
import gen_tile_engine as gen  # noqa: E402


def _regs(kind: str, text: str) -> set[int]:
    out = set()
    for a, b in re.findall(rf"\b{kind}\[(\d+):(\d+)\]", text):
        out.update(range(int(a), int(b) + 1))
    for a in re.findall(rf"\b{kind}(\d+)\b", text):
        out.add(int(a))
    return out


@pytest.mark.parametrize("partial", [False, True])
def test_engine_text(partial):
    lines = gen.engine(partial)
    labels = {m.group(1) for ln in lines for m in [re.match(r"(\.Lqs_\w+_%=):$", ln)] if m}
    used = {m for ln in lines for m in re.findall(r"\.Lqs_\w+_%=", ln) if not ln.endswith(":")}
    assert used <= labels, used - labels
    assert len(labels) == len([ln for ln in lines if ln.endswith(":")]), "a label is defined twice"
    # branch tables: NENT entries per bank, right behind s_getpc + one branch over them
    i = lines.index("s_getpc_b64 s[20:21]")
    assert lines[i + 1].startswith("s_branch ") and all(ln.startswith("s_branch ") for ln in lines[i + 2:i + 2 + 2 * gen.NENT])
    assert not lines[i + 2 + 2 * gen.NENT].startswith("s_branch ")
    # x0..x7 (v4..v35) are in/out operands of the asm statement (pinned registers), everything else it touches is a clobber
    clobber_v = {int(x) for x in re.findall(r'"v(\d+)"', gen.clobbers())}
    assert not clobber_v & set(range(4, 36)), "an operand register listed as a clobber"
    clobber_v |= set(range(4, 36))
    clobber_s = {int(x) for x in re.findall(r'"s(\d+)"', gen.clobbers())}
    for ln in lines:
        if ln.endswith(":"):
            continue
        body = re.sub(r"%\[\w+\]", "", ln)
        assert "{" not in ln and "}" not in ln and "|" not in ln, "asm dialect characters"
        assert _regs("v", body) <= clobber_v, (ln, _regs("v", body) - clobber_v)
        assert _regs("s", body) <= clobber_s, (ln, _regs("s", body) - clobber_s)
        if ln.startswith("v_") and not ln.startswith("v_cmp"):
            ops = body.split(None, 1)[1].split(",")
            srcs = [o.strip().lstrip("-") for o in ops[1:]]
            sgpr_srcs = {o for o in srcs if re.match(r"s(\[|\d)", o)}
            assert len(sgpr_srcs) <= 1, f"more than one SGPR source (constant-bus limit): {ln}"
            for o in srcs + [ops[0].strip()]:          # 64-bit operands are even-aligned pairs
                m = re.match(r"[vs]\[(\d+):(\d+)\]", o)
                if m and int(m.group(2)) - int(m.group(1)) == 1:
                    assert int(m.group(1)) % 2 == 0, ln
        if ln.startswith("s_load_dwordx"):
            m = re.match(r"s_load_dwordx(\d+) s\[(\d+):(\d+)\]", ln)
            assert int(m.group(3)) - int(m.group(2)) + 1 == int(m.group(1)) and int(m.group(2)) % 4 == 0, ln
    assert 32 not in clobber_s, "s32 is the ABI stack pointer: keep the engine off it"
    assert max(clobber_s) <= 95 and max(clobber_v) <= 61


def test_every_case_ends_with_a_dispatch_and_the_committed_header_is_current():
    lines = gen.engine(False)
    cases = gen.gate_cases()
    for bank in "AB":
        for e, (name, _) in cases.items():
            start = lines.index(f".Lqs_{name}_{bank}_%=:")
            nxt = next(i for i in range(start + 1, len(lines)) if lines[i].endswith(":"))
            assert lines[nxt - 1] == f"s_branch .Lqs_top_{'B' if bank == 'A' else 'A'}_%=", (name, bank, lines[nxt - 1])
    header = (ROOT / "quantum_simulations_amd" / "csrc" / "tile_engine_gen.h").read_text()
    assert gen.c_string(lines).replace("\n", " \\\n") in header, "tile_engine_gen.h is stale: run make in csrc/"
    for name, value in gen.OPC.items():
        assert f"#define QS_ENT_{name} {value}\n" in header


def test_no_scalar_load_in_flight_when_the_statement_ends():
    """Every record is fetched one ahead, so a fetch is in flight at END; the compiler treats the bank registers as
    free after the statement -- a fetch landing later would overwrite its values (seen as a rare wrong result with
    OPC_END_DIRECT before the wait was there).  Both exits must wait for lgkmcnt(0) after their label."""
    for partial in (False, True):
        lines = gen.engine(partial)
        for label in (".Lqs_end_%=:", ".Lqs_end_direct_%=:"):
            at = lines.index(label)
            nxt = next((i for i in range(at + 1, len(lines)) if lines[i].endswith(":")), len(lines))
            assert "s_waitcnt lgkmcnt(0)" in lines[at:nxt], label
        assert lines[-1] == "s_waitcnt lgkmcnt(0)" or "s_barrier" in lines[-2:], lines[-3:]


# ---- control-flow walk of the engine -----------------------------------------------------------
# The engine's contract with the compiler around the asm statement: (1) no scalar load in flight at either exit (the
# OPC_END_DIRECT bug of round 2), (2) EXEC back at all-ones at the exits after every hand-narrowing (predicated gates),
# (3) no s_barrier and no LDS traffic of a register-group change under a narrowed EXEC.  Instead of grepping the last
# lines, the checks below follow EVERY path: labels, s_branch, s_cbranch_scc0/1 and the direct-threaded dispatch
# (s_add_u32 s24, s20|s22 ... s_setpc_b64 = any entry of that bank's branch table).
def _cfg(lines):
    label_at = {ln[:-1]: i for i, ln in enumerate(lines) if ln.endswith(":")}
    t0 = lines.index("s_getpc_b64 s[20:21]") + 2
    table = {"A": list(range(t0, t0 + gen.NENT)), "B": list(range(t0 + gen.NENT, t0 + 2 * gen.NENT))}
    # second dispatch of a predicated gate (offset from header dword 3, in s18): the host writes gate CASES only there
    # (serialize_pass; tests/tile_interpreter.py asserts it on every planned image), never a group / end / predicate entry
    gate_entries = sorted(gen.gate_cases())
    succ = {}
    bank_of_dispatch = None
    for i, ln in enumerate(lines):
        nxt = [i + 1] if i + 1 < len(lines) else ["EXIT"]
        m = re.match(r"s_add_u32 s24, s(20|22), (\S+)", ln)
        if m:
            bank_of_dispatch = ("A" if m.group(1) == "20" else "B", m.group(2) == "s18")
        if ln.startswith("s_branch "):
            succ[i] = [label_at[ln.split()[1]]]
        elif ln.startswith("s_cbranch_"):
            succ[i] = [label_at[ln.split()[1]]] + nxt
        elif ln.startswith("s_setpc_b64"):
            assert bank_of_dispatch is not None, "a dispatch without its table base"
            bank, second = bank_of_dispatch
            succ[i] = [table[bank][e] for e in gate_entries] if second else table[bank]
            bank_of_dispatch = None
        else:
            succ[i] = nxt
    return succ


def _flow(lines, succ, start_state, transfer):
    """forward may-analysis: state = frozenset of abstract values; returns the state IN FRONT of every instruction
    (and of 'EXIT')."""
    state_in = {0: frozenset(start_state)}
    work = [0]
    while work:
        i = work.pop()
        out = frozenset(v2 for v in state_in[i] for v2 in transfer(lines[i], v))
        for j in succ[i]:
            merged = state_in.get(j, frozenset()) | out
            if merged != state_in.get(j):
                state_in[j] = merged
                if j != "EXIT":
                    work.append(j)
    return state_in


@pytest.mark.parametrize("partial", [False, True])
def test_control_flow_every_path_to_the_exits(partial):
    lines = gen.engine(partial)
    succ = _cfg(lines)

    # (1) scalar loads in flight: True after s_load, False after a wait for lgkmcnt(0)
    def smem(ln, inflight):
        if ln.startswith("s_load_dword"):
            return [True]
        if ln.startswith("s_waitcnt") and "lgkmcnt(0)" in ln:
            return [False]
        return [inflight]
    st = _flow(lines, succ, [False], smem)
    assert "EXIT" in st, "no path reaches the end of the statement"
    assert st["EXIT"] == frozenset([False]), "a path leaves the statement with a scalar load in flight"
    # every instruction is reachable except the entries of unused table slots' targets (none are dead code here)
    dead = [i for i in range(len(lines)) if i not in st and not lines[i].endswith(":")]
    assert not dead, [lines[i] for i in dead[:5]]
    # a bank is only read after its fetch has landed: no VALU / SALU instruction that names a bank register
    # (s36..s83) may execute with a load possibly in flight -- except the fetch of the NEXT record itself, which is
    # issued right after the wait and targets the other bank
    for i, ln in enumerate(lines):
        if i in st and True in st[i] and not ln.endswith(":") and not ln.startswith(("s_load_dword", "s_waitcnt")):
            regs = _regs("s", re.sub(r"%\[\w+\]", "", ln))
            if regs & set(range(36, 84)):
                # allowed: reading the CURRENT bank while the look-ahead fetch of the other bank is in flight --
                # the current bank was waited for at its top_ label.  What must not happen is a read of the bank
                # that is being fetched: checked per bank below.
                pass

    # per bank: a register of bank X (A = s36..51, B = s52..67, E = s68..83) must not be read while a load INTO X may
    # be in flight
    banks = {"A": set(range(36, 52)), "B": set(range(52, 68)), "E": set(range(68, 84))}
    for name, regs_of in banks.items():
        def pending(ln, inflight, regs_of=regs_of):
            m = re.match(r"s_load_dwordx\d+ s\[(\d+):(\d+)\]", ln)
            if m and set(range(int(m.group(1)), int(m.group(2)) + 1)) & regs_of:
                return [True]
            if ln.startswith("s_waitcnt") and "lgkmcnt(0)" in ln:
                return [False]
            return [inflight]
        stb = _flow(lines, succ, [False], pending)
        for i, ln in enumerate(lines):
            if ln.endswith(":") or ln.startswith(("s_load_dword", "s_waitcnt")) or i not in stb or True not in stb[i]:
                continue
            body = re.sub(r"%\[\w+\]", "", ln)
            ops = body.split(None, 1)[1] if " " in body else ""
            srcs = ops.split(",")[1:] if not ln.startswith(("s_cmp", "s_branch", "s_cbranch", "s_setpc", "s_barrier", "s_nop", "ds_write")) else [ops]
            read = set().union(*[_regs("s", o) for o in srcs]) if srcs else set()
            assert not (read & regs_of), f"bank {name} read while its fetch may be in flight: {ln}"

    # (2) + (3) EXEC: FULL (all ones), LIVE (tiles smaller than 8 x blockDim: s[28:29]), NARROW (hand-narrowed)
    def exec_state(ln, v):
        if ln == "s_mov_b64 exec, -1":
            return ["FULL"]
        if ln == "s_mov_b64 exec, s[28:29]":
            return ["LIVE"]
        if re.match(r"s_and_b64 exec, exec,", ln):
            return ["NARROW"]
        assert not re.match(r"s_\w+ exec\b", ln), f"unmodelled write to EXEC: {ln}"
        return [v]
    se = _flow(lines, succ, ["FULL"], exec_state)
    assert se["EXIT"] == frozenset(["FULL"]), f"EXEC at the end of the statement: {set(se['EXIT'])}"
    for i, ln in enumerate(lines):
        if i not in se:
            continue
        if ln == "s_barrier":
            assert se[i] == frozenset(["FULL"]), f"s_barrier under EXEC {set(se[i])} (line {i})"
        if ln.startswith(("ds_write_b128", "ds_read_b128")):
            assert "NARROW" not in se[i], f"register-group LDS traffic under a narrowed EXEC (line {i}): {ln}"
            if partial:
                assert se[i] == frozenset(["LIVE"]), f"partial tiles: only live lanes own a register block (line {i}: {set(se[i])})"
    # every gate body starts from a defined EXEC: FULL (or narrowed by ITS OWN predicate entry), never a leftover
    for bank in "AB":
        at = lines.index(f".Lqs_top_{bank}_%=:")
        assert lines[at + 1] == "s_mov_b64 exec, -1", "the top of a record restores EXEC before anything else"


def test_control_flow_checker_catches_the_round_2_bug():
    """The same walk on an engine with the wait at END_DIRECT removed must fail (the checker is not vacuous)."""
    lines = gen.engine(False)
    at = lines.index(".Lqs_end_direct_%=:")
    broken = lines[:at + 1] + [ln for ln in lines[at + 1:] if ln != "s_waitcnt lgkmcnt(0)"]
    succ = _cfg(broken)

    def smem(ln, inflight):
        if ln.startswith("s_load_dword"):
            return [True]
        if ln.startswith("s_waitcnt") and "lgkmcnt(0)" in ln:
            return [False]
        return [inflight]
    st = _flow(broken, succ, [False], smem)
    assert True in st["EXIT"]
