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
