#!/usr/bin/env python3
"""The layout search on the secondary single-GPU workloads (round 4): GHZ+QFT, Clifford+T depth 60, random 1q+CX depth
40 at n qubits through SingleGpuEngine with layout = identity / search (+ tuning on the device): passes, ms per step (HIP
events over 5 steps), and for GHZ+QFT the closed-form error through the layout.   python tools/layout_workloads.py [n]"""
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from quantum_simulations_amd import circuits as gen  # noqa: E402
from quantum_simulations_amd.runner.engine import SingleGpuEngine  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
for name, cd in (("ghz_qft", gen.generate_ghz_qft(n)), ("clifford_t_d60", gen.random_clifford_t_circuit(n, depth=60)),
                 ("random_1q_cx_d40_seed7", gen.random_1q_cx_circuit(n, depth=40, seed=7))):
    rec = {"circuit": name, "n_qubits": n, "gates": len(cd["gates"])}
    for layout in ("identity", "search"):
        eng = SingleGpuEngine(n, layout=layout, tune_on_device=layout == "search")
        eng.init_zero_state()
        plan = eng.plan(cd)
        eng.execute(plan)
        err = eng.state.max_abs_err_closed_form("ghz_qft", n, 0, eng.l2p) if name == "ghz_qft" else None
        eng.barrier()
        eng.state.time_begin()
        for _ in range(5):
            eng.execute(plan)
        ms = eng.state.time_end() / 5
        rec[layout] = {"ms_per_step": round(ms, 3), "hbm_passes": eng.last_passes, "gate_apps_per_s": round(len(cd["gates"]) / ms * 1e3, 1),
                       "norm2": eng.norm2()}
        if err is not None:
            rec[layout]["max_abs_err_vs_closed_form_first_execution"] = err
        info = getattr(plan, "layout_info", None)
        if info:
            rec[layout]["layout_search"] = info
        eng.close()
    rec["speedup"] = round(rec["identity"]["ms_per_step"] / rec["search"]["ms_per_step"], 4)
    print(json.dumps(rec), flush=True)
