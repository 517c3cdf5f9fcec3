#!/usr/bin/env python3
"""Offline (no GPU): passes the fused-pass builder needs when consecutive tiles must share >= s of their 8 high bits
(QSIM_PLAN_ANCHOR = s, probe build) -- the sliding-window / anchored-tile question of VERDICT r03 item 3.
    python tools/anchor_pass_counts.py [n ...]      one child process per (s, n): the knob is read once per process"""
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
LIB = os.environ.get("QSIM_LIBRARY") or str(ROOT / "quantum_simulations_amd" / "libqsim_hip_probes.so")

CHILD = r"""
import ctypes as C, json, sys
sys.path.insert(0, %r)
from quantum_simulations_amd import _lib
from quantum_simulations_amd.circuits import random_1q_cx_circuit, random_clifford_t_circuit, generate_ghz_qft
from quantum_simulations_amd.circuit.io import levelize, validate_circuit_dict
from quantum_simulations_amd.circuit.fusion import batch_levels
from quantum_simulations_amd.kernel.device import pack_ops
n = int(sys.argv[1])
lib = _lib.load()
def count(cd):
    total = 0
    for p in batch_levels(levelize(validate_circuit_dict(cd)), n):      # what SingleGpuEngine.plan hands to qsim_apply_ops
        nq, qubits, mats = pack_ops(p["local_ops"])
        k = C.c_int32()
        _lib.check(lib.qsim_plan_ops(n, len(nq), nq.ctypes.data_as(C.c_void_p), qubits.ctypes.data_as(C.c_void_p),
                                     mats.ctypes.data_as(C.c_void_p), None, 0, C.byref(k)))
        total += k.value
    return total
out = {}
out["bench"] = count(random_1q_cx_circuit(n, depth=40))
out["random8"] = [count(random_1q_cx_circuit(n, depth=40, seed=s)) for s in range(1, 9)]
out["clifford_t6"] = [count(random_clifford_t_circuit(n, depth=60, seed=s)) for s in range(1, 7)]
out["ghz_qft"] = count(generate_ghz_qft(n))
print(json.dumps(out))
"""


def main():
    ns = [int(a) for a in sys.argv[1:]] or [28, 30]
    for n in ns:
        for s in (0, 1, 2, 3, 4, 5, 6):
            env = dict(os.environ, QSIM_LIBRARY=LIB, QSIM_PLAN_ANCHOR=str(s), PYTHONPATH=str(ROOT))
            r = subprocess.run([sys.executable, "-c", CHILD % str(ROOT), str(n)], env=env, capture_output=True, text=True)
            if r.returncode:
                print(r.stderr, file=sys.stderr)
                raise SystemExit(1)
            d = json.loads(r.stdout.strip().splitlines()[-1])
            print(f"n={n} shared>={s}: bench {d['bench']:3d}   8 random 1q+CX {sum(d['random8']):4d} {d['random8']}   "
                  f"6 Clifford+T {sum(d['clifford_t6']):4d}   GHZ+QFT {d['ghz_qft']}", flush=True)


if __name__ == "__main__":
    main()
