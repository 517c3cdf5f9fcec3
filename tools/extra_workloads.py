#!/usr/bin/env python3
"""Secondary single-GPU workloads (not the bench headline): GHZ+QFT (config 5's circuit at one
GPU's share) and the Clifford+T circuit of config 4, fused, timed with HIP events."""
import json
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from quantum_simulations_amd import circuits as gen  # noqa: E402
from quantum_simulations_amd.circuit.io import validate_circuit_dict  # noqa: E402
from quantum_simulations_amd.kernel.device import DeviceChunk  # noqa: E402
from quantum_simulations_amd.runner.engine import gate_ops  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = DeviceChunk.empty(n)
for name, cd in (("ghz_qft", gen.generate_ghz_qft(n)), ("clifford_t_d60", gen.random_clifford_t_circuit(n, depth=60)),
                 ("random_1q_cx_d40", gen.random_1q_cx_circuit(n, depth=40))):
    ops = gate_ops(validate_circuit_dict(cd))
    dev.init_zero(True)
    dev.apply_ops(ops)
    dev.sync()
    rec = {"circuit": name, "n_qubits": n, "gates": len(ops)}
    for fused in (True, False):
        dev.init_zero(True)
        dev.sync()
        dev.time_begin()
        passes = dev.apply_ops(ops, fused=fused)
        ms = dev.time_end()
        rec["fused" if fused else "per_gate"] = {"ms": round(ms, 2), "hbm_passes": passes,
                                                 "gate_apps_per_s": round(len(ops) / ms * 1e3, 1)}
    if name == "ghz_qft":
        rec["max_abs_err_vs_closed_form"] = dev.max_abs_err_closed_form("ghz_qft", n)
    print(json.dumps(rec), flush=True)
dev.close()
