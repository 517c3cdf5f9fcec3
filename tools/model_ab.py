#!/usr/bin/env python3
"""A/B of tile-cost models for the layout search on one device: for every model file given, the bench circuit at n qubits
through SingleGpuEngine(layout="search", tune_on_device) -- passes, model prediction, measured ms per step (best of the
timed finalists and a fresh 5-step timing).   python tools/model_ab.py n model.json [model.json ...]"""
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from quantum_simulations_amd import circuits as gen  # noqa: E402
from quantum_simulations_amd.runner import tile_layout  # noqa: E402
from quantum_simulations_amd.runner.engine import SingleGpuEngine  # noqa: E402

n = int(sys.argv[1])
cd = gen.random_1q_cx_circuit(n, depth=40)
eng = SingleGpuEngine(n, layout="search", tune_on_device=True)
for rep in range(2):
    for path in sys.argv[2:] + ["identity"]:
        if path == "identity":
            eng.layout_mode = "identity"
        else:
            eng.layout_mode = "search"
            tile_layout._MODEL_PATH = Path(path)
            tile_layout._models = None
        eng.init_zero_state()
        plan = eng.plan(cd)
        eng.execute(plan)
        eng.barrier()
        eng.state.time_begin()
        for _ in range(5):
            eng.execute(plan)
        ms = eng.state.time_end() / 5
        info = getattr(plan, "layout_info", {})
        print(json.dumps({"model": Path(path).name, "n": n, "ms_per_step": round(ms, 3), "passes": eng.last_passes,
                          "model_ms": info.get("model_ms"), "tuned": info.get("tuned_on_device_ms")}), flush=True)
eng.close()
