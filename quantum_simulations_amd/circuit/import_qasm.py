"""OpenQASM 2.0 front-end: the part of QASMBench that the reference's gate set can express
(inputs of that shape ship under
v3_hisvsim_spark/hisvsim_repo/QASMBench/ in the reference -- its Python path never reads them).

`qasm_to_dict(text)` -> circuit dict of the circuit contract (wenbo_engine/docs/circuit_contract.md).
The contract's gates are H X Y Z S T RY R(k) G(p) CNOT SWAP CZ CY CR(k) CU -- there is NO general 1q
rotation, so only what maps EXACTLY is accepted:

    h x y z s t ry cx cz cy swap id                 one gate each
    sdg = Z S, tdg = Z S T                          (diagonal, exact)
    u1 / p (lambda), cu1 / cp (lambda)              when lambda = 2 pi m / 2^K, K <= 48: a product of R(k) / CR(k)
    ccx, cswap                                      the standard 15-gate Clifford+T decomposition
    user `gate` definitions                         expanded in place
    barrier, measure (terminal), creg               dropped

Everything else (rx rz u2 u3 rzz ryy crz ch reset if(...) ...) raises ValueError("unsupported gate ...")
like validate_circuit_dict does for unknown names.  Parity: the importer has no counterpart in the
reference, so there is no reference fixture for it -- "parity unpinned"; the tests check it against
explicit matrices on small registers.  Qubit q[i] of the first register is qubit i (bit i of the
amplitude index: qiskit's and this engine's little-endian convention); further registers follow.
"""
from __future__ import annotations

import math
import re

_SIMPLE = {"h": "H", "x": "X", "y": "Y", "z": "Z", "s": "S", "t": "T", "cx": "CNOT", "cnot": "CNOT",
           "cz": "CZ", "cy": "CY", "swap": "SWAP"}
_ARITY = {"h": 1, "x": 1, "y": 1, "z": 1, "s": 1, "t": 1, "sdg": 1, "tdg": 1, "id": 1, "ry": 1, "u1": 1, "p": 1,
          "cx": 2, "cnot": 2, "cz": 2, "cy": 2, "swap": 2, "cu1": 2, "cp": 2, "ccx": 3, "cswap": 3}
_NPARAMS = {"ry": 1, "u1": 1, "p": 1, "cu1": 1, "cp": 1}
_MAX_K = 48


def _unsupported(name: str, why: str = "") -> ValueError:
    return ValueError(f"unsupported gate '{name}'" + (f": {why}" if why else "")
                      + " (the circuit contract has no general 1-qubit rotation; see import_qasm.py)")


# ---- parameter expressions: numbers, pi, + - * / ^, parentheses, unary minus, a few functions ----
_TOKEN = re.compile(r"\s*(?:(\d+\.?\d*(?:[eE][+-]?\d+)?|\.\d+(?:[eE][+-]?\d+)?)|([A-Za-z_][A-Za-z_0-9]*)|(.))")


def _eval_expr(text: str, env: dict) -> float:
    tokens = [(m.group(1), m.group(2), m.group(3)) for m in _TOKEN.finditer(text) if m.group(0).strip()]
    pos = 0

    def peek():
        return tokens[pos] if pos < len(tokens) else (None, None, None)

    def take():
        nonlocal pos
        pos += 1
        return tokens[pos - 1]

    def atom() -> float:
        num, name, sym = take()
        if num is not None:
            return float(num)
        if name is not None:
            if name == "pi":
                return math.pi
            if name in env:
                return env[name]
            if name in ("sin", "cos", "tan", "exp", "ln", "sqrt"):
                if take()[2] != "(":
                    raise ValueError(f"bad expression {text!r}")
                v = expr()
                if take()[2] != ")":
                    raise ValueError(f"bad expression {text!r}")
                return {"sin": math.sin, "cos": math.cos, "tan": math.tan, "exp": math.exp, "ln": math.log,
                        "sqrt": math.sqrt}[name](v)
            raise ValueError(f"unknown identifier {name!r} in expression {text!r}")
        if sym == "(":
            v = expr()
            if take()[2] != ")":
                raise ValueError(f"bad expression {text!r}")
            return v
        if sym == "-":
            return -power()
        if sym == "+":
            return power()
        raise ValueError(f"bad expression {text!r}")

    def power() -> float:
        base = atom()
        if peek()[2] == "^":
            take()
            return base ** power()
        return base

    def term() -> float:
        v = power()
        while peek()[2] in ("*", "/"):
            op = take()[2]
            r = power()
            v = v * r if op == "*" else v / r
        return v

    def expr() -> float:
        v = term()
        while peek()[2] in ("+", "-"):
            op = take()[2]
            r = term()
            v = v + r if op == "+" else v - r
        return v

    out = expr()
    if pos != len(tokens):
        raise ValueError(f"bad expression {text!r}")
    return out


def _phase_powers(name: str, lam: float) -> list[int]:
    """lambda = 2 pi m / 2^K exactly (to the precision a double angle carries) -> the k's with
    diag(1, e^{i lambda}) = prod R(k): the smallest K <= 48 for which lambda / 2 pi * 2^K is an integer."""
    turns = (lam / (2.0 * math.pi)) % 1.0
    for K in range(0, _MAX_K + 1):
        x = turns * (1 << K)
        slack = 4e-15 * x                      # what the rounding of lam / (2 pi) can have moved x by
        if slack > 1e-3:
            break
        m = round(x)
        if abs(x - m) <= max(1e-9, slack):
            m %= 1 << K
            return [k for k in range(1, K + 1) if (m >> (K - k)) & 1]
    raise _unsupported(name, f"angle {lam!r} is not a multiple of 2 pi / 2^k (k <= {_MAX_K})")


def _emit_builtin(name: str, params: list[float], q: list[int], out: list) -> None:
    def g(gate, qubits, **p):
        out.append({"qubits": list(qubits), "gate": gate, "params": p})

    if name in _SIMPLE:
        g(_SIMPLE[name], q)
    elif name == "id":
        pass
    elif name == "ry":
        g("RY", q, theta=float(params[0]))
    elif name == "sdg":
        g("Z", q)
        g("S", q)
    elif name == "tdg":
        g("Z", q)
        g("S", q)
        g("T", q)
    elif name in ("u1", "p"):
        for k in _phase_powers(name, params[0]):
            g("R", q, k=k)
    elif name in ("cu1", "cp"):
        for k in _phase_powers(name, params[0]):
            g("CR", q, k=k)
    elif name == "ccx":
        a, b, c = q
        for nm, qs in (("h", [c]), ("cx", [b, c]), ("tdg", [c]), ("cx", [a, c]), ("t", [c]), ("cx", [b, c]),
                       ("tdg", [c]), ("cx", [a, c]), ("t", [b]), ("t", [c]), ("h", [c]), ("cx", [a, b]),
                       ("t", [a]), ("tdg", [b]), ("cx", [a, b])):
            _emit_builtin(nm, [], qs, out)
    elif name == "cswap":
        a, b, c = q
        _emit_builtin("cx", [], [c, b], out)
        _emit_builtin("ccx", [], [a, b, c], out)
        _emit_builtin("cx", [], [c, b], out)
    else:
        raise _unsupported(name)


_STMT_GATE = re.compile(r"^([A-Za-z_][A-Za-z_0-9]*)\s*(?:\((.*)\))?\s*(.*)$", re.S)


def _split_args(text: str) -> list[str]:
    parts, depth, cur = [], 0, ""
    for ch in text:
        if ch == "(":
            depth += 1
        elif ch == ")":
            depth -= 1
        if ch == "," and depth == 0:
            parts.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        parts.append(cur.strip())
    return parts


def qasm_to_dict(text: str) -> dict:
    text = re.sub(r"//[^\n]*", "", text)
    # user gate definitions: gate name(params) qargs { body }
    macros: dict[str, tuple[list[str], list[str], list[str]]] = {}

    def grab_gate(m):
        head, body = m.group(1).strip(), m.group(2)
        hm = re.match(r"^([A-Za-z_][A-Za-z_0-9]*)\s*(?:\(([^)]*)\))?\s*(.*)$", head, re.S)
        name = hm.group(1)
        params = [p.strip() for p in (hm.group(2) or "").split(",") if p.strip()]
        qargs = [a.strip() for a in hm.group(3).split(",") if a.strip()]
        macros[name] = (params, qargs, [s.strip() for s in body.split(";") if s.strip()])
        return ""

    text = re.sub(r"\bgate\s+([^{]*)\{([^}]*)\}", grab_gate, text)
    if re.search(r"\bopaque\b", text):
        raise _unsupported("opaque")
    regs: dict[str, tuple[int, int]] = {}
    n_qubits = 0
    gates: list = []
    measured: set[int] = set()

    def resolve(arg: str) -> list[int]:
        m = re.match(r"^([A-Za-z_][A-Za-z_0-9]*)\s*(?:\[\s*(\d+)\s*\])?$", arg)
        if not m or m.group(1) not in regs:
            raise ValueError(f"unknown qubit argument {arg!r}")
        off, size = regs[m.group(1)]
        if m.group(2) is None:
            return [off + i for i in range(size)]
        i = int(m.group(2))
        if i >= size:
            raise ValueError(f"qubit index out of range in {arg!r}")
        return [off + i]

    def apply(name: str, params: list[float], qubits: list[int], depth: int = 0) -> None:
        if depth > 64:
            raise ValueError("gate definitions nest too deeply")
        if any(q in measured for q in qubits):
            raise _unsupported(name, "a gate after a measurement of the same qubit (mid-circuit measurement)")
        if len(set(qubits)) != len(qubits):
            raise ValueError(f"gate {name!r} names a qubit twice")
        if name in macros:
            pnames, qnames, body = macros[name]
            if len(pnames) != len(params) or len(qnames) != len(qubits):
                raise ValueError(f"gate {name!r}: wrong number of parameters or qubits")
            env = dict(zip(pnames, params))
            qenv = dict(zip(qnames, qubits))
            for stmt in body:
                sm = _STMT_GATE.match(stmt)
                sub = sm.group(1)
                if sub == "barrier":
                    continue
                sp = [_eval_expr(e, env) for e in _split_args(sm.group(2) or "")]
                sq = []
                for a in _split_args(sm.group(3)):
                    if a not in qenv:
                        raise ValueError(f"gate {name!r}: unknown qubit {a!r} in its body")
                    sq.append(qenv[a])
                apply(sub, sp, sq, depth + 1)
            return
        if name not in _ARITY:
            raise _unsupported(name)
        if len(qubits) != _ARITY[name] or len(params) != _NPARAMS.get(name, 0):
            raise ValueError(f"gate {name!r}: wrong number of parameters or qubits")
        _emit_builtin(name, params, qubits, gates)

    for raw in text.split(";"):
        stmt = raw.strip()
        if not stmt:
            continue
        if stmt.startswith("OPENQASM"):
            if not re.match(r"^OPENQASM\s+2(\.\d+)?$", stmt):
                raise ValueError(f"only OpenQASM 2 is read, got {stmt!r}")
            continue
        if stmt.startswith("include"):
            continue
        m = re.match(r"^(qreg|creg)\s+([A-Za-z_][A-Za-z_0-9]*)\s*\[\s*(\d+)\s*\]$", stmt)
        if m:
            if m.group(1) == "qreg":
                regs[m.group(2)] = (n_qubits, int(m.group(3)))
                n_qubits += int(m.group(3))
            continue
        if stmt.startswith("barrier"):
            continue
        if stmt.startswith("measure"):
            mm = re.match(r"^measure\s+(.+?)\s*->\s*(.+)$", stmt, re.S)
            if not mm:
                raise ValueError(f"bad measure statement {stmt!r}")
            measured.update(resolve(mm.group(1).strip()))
            continue
        if stmt.startswith("reset") or stmt.startswith("if"):
            raise _unsupported(stmt.split()[0].split("(")[0], "no classical control or reset in a statevector run")
        sm = _STMT_GATE.match(stmt)
        if not sm:
            raise ValueError(f"cannot parse {stmt!r}")
        name = sm.group(1)
        params = [_eval_expr(e, {}) for e in _split_args(sm.group(2) or "")]
        args = [resolve(a) for a in _split_args(sm.group(3))]
        if not args:
            raise ValueError(f"gate {name!r} without qubits")
        width = max(len(a) for a in args)
        if any(len(a) not in (1, width) for a in args):
            raise ValueError(f"register sizes differ in {stmt!r}")
        for i in range(width):                      # register broadcast
            apply(name, params, [a[i] if len(a) > 1 else a[0] for a in args])
    if n_qubits == 0:
        raise ValueError("no qreg declared")
    return {"number_of_qubits": n_qubits, "gates": gates}


def load_qasm(path) -> dict:
    with open(path) as f:
        return qasm_to_dict(f.read())
