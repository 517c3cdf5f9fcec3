"""OpenQASM 2.0 front-end: the part of QASMBench that the reference's gate set can express
(inputs of that shape ship under
v3_hisvsim_spark/hisvsim_repo/QASMBench/ in the reference -- its Python path never reads them).

`qasm_to_dict(text)` -> circuit dict of the circuit contract (wenbo_engine/docs/circuit_contract.md).
The contract's gates are H X Y Z S T RY R(k) G(p) CNOT SWAP CZ CY CR(k) CU.  What maps EXACTLY:

    h x y z s t ry cx cz cy swap id                 one gate each
    sdg = Z S, tdg = Z S T                          (diagonal, exact)
    u1 / p (lambda), cu1 / cp (lambda)              lambda = 2 pi m / 2^K, K <= 48: a product of R(k) / CR(k)
    cu1 / cp (any lambda), crz, crx, cry, cu3, ch,  the contract's CU with the 2x2 block as its `U` (exact)
    csx, cu(theta, phi, lambda, gamma)
    ccx, cswap                                      the standard 15-gate Clifford+T decomposition
    c3x                                             two ccx and five CU with sqrt(X) / its fourth root (exact)
    user `gate` definitions                         expanded in place
    barrier, measure (terminal), creg               dropped

and, exact UP TO A GLOBAL PHASE of the whole state (the contract has RY(theta) for every angle, and a rotation about any
other axis is RY between Cliffords: RX(a) = S^dagger RY(a) S, RZ(a) = (S H)^dagger RY(a) (S H) -- runs of 1q gates are fused
into one 2x2 before they reach the device, so the extra Cliffords cost nothing there):

    rx rz sx sxdg                                   the SU(2) rotation (qelib1's rz(a) = u1(a) = e^{i a / 2} RZ(a))
    u3(t, p, l) = e^{i (p + l) / 2} RZ(p) RY(t) RZ(l),  u2(p, l) = u3(pi / 2, p, l),  u / U = u3
    u1 / p with an angle that is no dyadic fraction of 2 pi: RZ(lambda) (phase e^{i lambda / 2} dropped)
    rzz(a) = cx; RZ(a) on the target; cx = exp(-i a / 2 Z x Z); rxx, ryy likewise (qelib1's forms carry a phase e^{i a / 2})

A statevector differs from qiskit's by that one unit-modulus factor; probabilities, expectation values and every
controlled use are unaffected.  Everything else (reset, if(...), opaque, mid-circuit measurement ...) raises
ValueError("unsupported gate ...") like validate_circuit_dict does for unknown names.  Parity: the importer has no counterpart in the
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
          "rx": 1, "rz": 1, "u2": 1, "u3": 1, "u": 1, "U": 1, "sx": 1, "sxdg": 1, "csx": 2, "cu": 2, "c3x": 4,
          "cx": 2, "cnot": 2, "cz": 2, "cy": 2, "swap": 2, "cu1": 2, "cp": 2, "crz": 2, "crx": 2, "cry": 2, "cu3": 2, "ch": 2,
          "rzz": 2, "rxx": 2, "ryy": 2, "ccx": 3, "cswap": 3}
_NPARAMS = {"ry": 1, "u1": 1, "p": 1, "cu1": 1, "cp": 1, "rx": 1, "rz": 1, "u2": 2, "u3": 3, "u": 3, "U": 3, "crz": 1, "crx": 1,
            "cry": 1, "cu3": 3, "rzz": 1, "rxx": 1, "ryy": 1, "cu": 4}
_MAX_K = 48


def _unsupported(name: str, why: str = "") -> ValueError:
    return ValueError(f"unsupported gate '{name}'" + (f": {why}" if why else "") + " (see circuit/import_qasm.py for what is read)")


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


def _dyadic_or_none(name: str, lam: float):
    try:
        return _phase_powers(name, lam)
    except ValueError:
        return None


def u3_matrix(theta: float, phi: float, lam: float):
    """qelib1's u3 (OpenQASM 2.0 spec): [[cos t/2, -e^{i l} sin t/2], [e^{i p} sin t/2, e^{i (p + l)} cos t/2]]"""
    import numpy as np
    c, s_ = math.cos(theta / 2.0), math.sin(theta / 2.0)
    return np.array([[c, -complex(math.cos(lam), math.sin(lam)) * s_],
                     [complex(math.cos(phi), math.sin(phi)) * s_, complex(math.cos(phi + lam), math.sin(phi + lam)) * c]], dtype=np.complex128)


def _emit_builtin(name: str, params: list[float], q: list[int], out: list) -> None:
    def g(gate, qubits, **p):
        out.append({"qubits": list(qubits), "gate": gate, "params": p})

    def rz(qubit, a):          # diag(e^{-i a / 2}, e^{i a / 2}) = (S H)^dagger RY(a) (S H): circuit order H, S, RY, S^dagger, H
        g("H", [qubit]), g("S", [qubit]), g("RY", [qubit], theta=float(a)), g("Z", [qubit]), g("S", [qubit]), g("H", [qubit])

    def rx(qubit, a):          # S^dagger RY(a) S: circuit order S, RY, S^dagger
        g("S", [qubit]), g("RY", [qubit], theta=float(a)), g("Z", [qubit]), g("S", [qubit])

    def controlled(U):         # the contract's CU: |0><0| x I + |1><1| x U, control = q[0]
        g("CU", q, U=U, exponent=1)

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
        ks = _dyadic_or_none(name, params[0])
        if ks is not None:
            for k in ks:
                g("R", q, k=k)
        else:                                  # (any other angle: the rotation, its phase e^{i lambda / 2} dropped)
            rz(q[0], params[0])
    elif name in ("cu1", "cp"):
        ks = _dyadic_or_none(name, params[0])
        if ks is not None:
            for k in ks:
                g("CR", q, k=k)
        else:
            controlled(u3_matrix(0.0, 0.0, params[0]))
    elif name == "rz":
        rz(q[0], params[0])
    elif name == "rx":
        rx(q[0], params[0])
    elif name in ("sx", "sxdg"):               # sqrt(X) = e^{i pi / 4} RX(pi / 2), its inverse e^{-i pi / 4} RX(-pi / 2)
        rx(q[0], math.pi / 2.0 if name == "sx" else -math.pi / 2.0)
    elif name == "csx":                        # controlled sqrt(X): the phase matters under a control -> CU, exact
        import numpy as np
        controlled(np.array([[1 + 1j, 1 - 1j], [1 - 1j, 1 + 1j]], dtype=np.complex128) / 2.0)
    elif name == "cu":                         # qelib1's cu(theta, phi, lambda, gamma) = controlled e^{i gamma} u3
        controlled(complex(math.cos(params[3]), math.sin(params[3])) * u3_matrix(*params[:3]))
    elif name in ("u3", "u", "U", "u2"):
        theta, phi, lam = (math.pi / 2.0, params[0], params[1]) if name == "u2" else params
        rz(q[0], lam)
        g("RY", q, theta=float(theta))
        rz(q[0], phi)
    elif name == "crz":
        import numpy as np
        controlled(np.diag([complex(math.cos(params[0] / 2), -math.sin(params[0] / 2)), complex(math.cos(params[0] / 2), math.sin(params[0] / 2))]))
    elif name == "crx":
        controlled(u3_matrix(params[0], -math.pi / 2.0, math.pi / 2.0))
    elif name == "cry":
        controlled(u3_matrix(params[0], 0.0, 0.0))
    elif name == "cu3":
        controlled(u3_matrix(*params))
    elif name == "ch":
        import numpy as np
        controlled(np.array([[1, 1], [1, -1]], dtype=np.complex128) / math.sqrt(2.0))
    elif name == "rzz":                        # exp(-i a / 2 Z x Z)
        _emit_builtin("cx", [], q, out)
        rz(q[1], params[0])
        _emit_builtin("cx", [], q, out)
    elif name == "rxx":                        # exp(-i a / 2 X x X) = (H x H) rzz (H x H)
        g("H", [q[0]]), g("H", [q[1]])
        _emit_builtin("rzz", params, q, out)
        g("H", [q[0]]), g("H", [q[1]])
    elif name == "ryy":                        # exp(-i a / 2 Y x Y): rzz between RX(pi / 2) and RX(-pi / 2) on both qubits
        rx(q[0], math.pi / 2.0), rx(q[1], math.pi / 2.0)
        _emit_builtin("rzz", params, q, out)
        rx(q[0], -math.pi / 2.0), rx(q[1], -math.pi / 2.0)
    elif name == "ccx":
        a, b, c = q
        for nm, qs in (("h", [c]), ("cx", [b, c]), ("tdg", [c]), ("cx", [a, c]), ("t", [c]), ("cx", [b, c]),
                       ("tdg", [c]), ("cx", [a, c]), ("t", [b]), ("t", [c]), ("h", [c]), ("cx", [a, b]),
                       ("t", [a]), ("tdg", [b]), ("cx", [a, b])):
            _emit_builtin(nm, [], qs, out)
    elif name == "c3x":                        # three controls: V = sqrt(X) controlled by c, ccx(a, b -> c), V^dagger, ccx, V by (a, b)
        import numpy as np                     # (Barenco et al. lemma 7.5 with the last double-controlled V spelled out)
        a, b, c, t = q
        V = np.array([[1 + 1j, 1 - 1j], [1 - 1j, 1 + 1j]], dtype=np.complex128) / 2.0
        sq = np.linalg.eig(V)                  # sqrt(V) through its eigenvectors (H diagonalises it: eigenvalues 1 and i)
        W = (sq[1] * np.sqrt(sq[0].astype(np.complex128))) @ np.linalg.inv(sq[1])
        def cu(ctrl, U):
            out.append({"qubits": [ctrl, t], "gate": "CU", "params": {"U": U, "exponent": 1}})
        cu(c, V)
        _emit_builtin("ccx", [], [a, b, c], out)
        cu(c, V.conj().T)
        _emit_builtin("ccx", [], [a, b, c], out)
        cu(b, W), _emit_builtin("cx", [], [a, b], out), cu(b, W.conj().T), _emit_builtin("cx", [], [a, b], out), cu(a, W)
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
