"""OpenQASM 2.0 texts written by this build, in the families of the QASMBench inputs that ship with the reference
(v3_hisvsim_spark/hisvsim_repo/QASMBench/cluster/{bv_n14,adder_n10,qft_n15,qpe_n9}); the reference's own files are
not copied.  Shared by tests/test_import_qasm.py (CPU) and tests/test_gpu_qasm.py (GPU)."""
import math

HDR = 'OPENQASM 2.0;\ninclude "qelib1.inc";\n'


def bernstein_vazirani(n: int, secret: int) -> str:
    """n - 1 data qubits + one ancilla (the last): H wall, oracle = cx per secret bit, H wall."""
    m = n - 1
    lines = [HDR, f"qreg q[{n}];", f"creg c[{m}];", f"x q[{m}];", "h q;"]
    lines += [f"cx q[{i}],q[{m}];" for i in range(m) if (secret >> i) & 1]
    lines += [f"h q[{i}];" for i in range(m)]
    lines += [f"measure q[{i}] -> c[{i}];" for i in range(m)]
    return "\n".join(lines)


def ripple_adder(bits: int, a: int, b: int) -> str:
    """Cuccaro ripple-carry adder (the shape of QASMBench's adder): cin, a[bits], b[bits], cout;
    majority / unmaj as user gates with ccx.  b <- a + b, cout <- carry."""
    lines = [HDR, "gate majority a,b,c { cx c,b; cx c,a; ccx a,b,c; }",
             "gate unmaj a,b,c { ccx a,b,c; cx c,a; cx a,b; }",
             "qreg cin[1];", f"qreg a[{bits}];", f"qreg b[{bits}];", "qreg cout[1];", f"creg ans[{bits + 1}];"]
    lines += [f"x a[{i}];" for i in range(bits) if (a >> i) & 1]
    lines += [f"x b[{i}];" for i in range(bits) if (b >> i) & 1]
    lines.append("majority cin[0],b[0],a[0];")
    lines += [f"majority a[{i - 1}],b[{i}],a[{i}];" for i in range(1, bits)]
    lines.append(f"cx a[{bits - 1}],cout[0];")
    lines += [f"unmaj a[{i - 1}],b[{i}],a[{i}];" for i in range(bits - 1, 0, -1)]
    lines.append("unmaj cin[0],b[0],a[0];")
    lines += [f"measure b[{i}] -> ans[{i}];" for i in range(bits)] + [f"measure cout[0] -> ans[{bits}];"]
    return "\n".join(lines)


def qft_cu1(n: int, prepare: int) -> str:
    """|prepare> then the textbook QFT written with cu1(pi/2^d) (no final swaps), as in qft_n15."""
    lines = [HDR, f"qreg q[{n}];"] + [f"x q[{i}];" for i in range(n) if (prepare >> i) & 1]
    for j in reversed(range(n)):
        lines.append(f"h q[{j}];")
        lines += [f"cu1(pi/{1 << (j - k)}) q[{k}],q[{j}];" for k in reversed(range(j))]
    return "\n".join(lines)


def phase_estimation(t: int, numerator: int) -> str:
    """t counting qubits + one eigenstate qubit |1> of u1(2 pi numerator / 2^t); controlled powers as cu1 with
    dyadic angles, then the inverse QFT on the counting register: the register ends in |numerator> exactly."""
    lines = [HDR, f"qreg c[{t}];", "qreg e[1];", "x e[0];", "h c;"]
    for j in range(t):                       # counting qubit j controls U^(2^j): phase 2 pi numerator 2^j / 2^t
        num = (numerator << j) % (1 << t)
        if num:
            g = math.gcd(num, 1 << t)
            lines.append(f"cu1(2*pi*{num // g}/{(1 << t) // g}) c[{j}],e[0];")
    # inverse QFT on the bit-reversed convention: undo the qft_cu1 gate order with negated angles, then the
    # counting register reads the integer directly after the swap network
    for i in range(t // 2):
        lines.append(f"swap c[{i}],c[{t - 1 - i}];")
    for j in range(t):
        lines += [f"cu1(-pi/{1 << (j - k)}) c[{k}],c[{j}];" for k in range(j)]
        lines.append(f"h c[{j}];")
    return "\n".join(lines)


def ising_trotter(n: int, steps: int, seed: int = 3) -> str:
    """Trotterised transverse-field Ising chain in the shape of QASMBench's ising / qaoa inputs: layers of rzz on a ring,
    rx and rz with arbitrary angles on every qubit, a u3 and a crz thrown in -- the gates the importer maps through RY
    between Cliffords and through the contract's CU."""
    import random
    rnd = random.Random(seed)
    lines = [HDR, f"qreg q[{n}];", "h q;"]
    for _ in range(steps):
        lines += [f"rzz({rnd.uniform(-1.5, 1.5):.9f}) q[{i}],q[{(i + 1) % n}];" for i in range(n)]
        lines += [f"rx({rnd.uniform(-3, 3):.9f}) q[{i}];" for i in range(n)]
        lines += [f"rz({rnd.uniform(-3, 3):.9f}) q[{i}];" for i in range(0, n, 2)]
        a, b = rnd.sample(range(n), 2)
        lines.append(f"u3({rnd.uniform(0, 3):.9f},{rnd.uniform(-3, 3):.9f},{rnd.uniform(-3, 3):.9f}) q[{a}];")
        lines.append(f"crz({rnd.uniform(-3, 3):.9f}) q[{a}],q[{b}];")
    return "\n".join(lines)
