#!/usr/bin/env python3
"""One GPU: per-GPU COMPUTE time of rank 0's schedule of a circuit on N ranks under the staging methods -- "belady" (stage
sets first, every stage's gates planned alone; with the round-4 initial-layout search), "tiles" (stage boundaries and tile
passes planned together, runner/partition_plan.py) with the identity start layout, and "tiles" as the engine runs it (start
layout searched, local slots placed by the tile-cost model).  The op lists rank 0 would hand to the library -- recorded by a
planning twin of the engine, with the tiles the planner names -- run on ONE shard-sized chunk: same kernels and plans as on
the node; the exchanges are skipped (the data is meaningless, the timing is not; the slab stores of the fused re-layouts
are not part of it).
    python tools/shard_compute_probe.py [N_QUBITS N_RANKS [REPEATS [RANK]]]     (RANK: whose op lists; default 0; the last rank
                                                                               applies every gate with a global control)"""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from quantum_simulations_amd import circuits as gen  # noqa: E402
from quantum_simulations_amd.circuit.io import validate_circuit_dict  # noqa: E402
from quantum_simulations_amd.kernel.device import DeviceChunk, pack_ops  # noqa: E402
from quantum_simulations_amd.runner.distributed import DistributedEngine, PlanningBackend  # noqa: E402

n, world = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (32, 4)
repeats = int(sys.argv[3]) if len(sys.argv) > 3 else 8
rank = int(sys.argv[4]) if len(sys.argv) > 4 else 0
p = world.bit_length() - 1
k = n - p
chunk = DeviceChunk.empty(k)
chunk.init_random(3)
VARIANTS = (("belady + layout search (r04)", dict(staging_method="belady", layout="search")),
            ("tiles, identity start", dict(staging_method="tiles", layout="identity")),
            ("tiles, searched start + slots placed", dict(staging_method="tiles", layout="search")))
for name, cd, reps in (("random 1q+CX depth 40", gen.random_1q_cx_circuit(n, depth=40), repeats), ("Clifford+T depth 60", gen.random_clifford_t_circuit(n, depth=60), 1),
                       ("GHZ+QFT", gen.generate_ghz_qft(n), 1)):
    cd = validate_circuit_dict(cd)
    for label, kw in VARIANTS:
        eng = DistributedEngine(n, world, rank, backend=PlanningBackend(k), init_process_group=False, **kw)
        eng.init_zero_state()
        t0 = time.perf_counter()
        plan = eng.plan(cd, repeats=reps)
        plan_s = time.perf_counter() - t0
        per_exec = []
        for _ in range(reps):
            eng.backend.record = []
            eng.relayout_log = []
            eng.execute(plan)
            lists = [(pack_ops(ops), None if tiles is None else np.array(tiles, dtype=np.uint64)) for ops, tiles in eng.backend.record]
            relayouts = list(eng.relayout_log)

            def run():
                total = 0
                for ops, tiles in lists:
                    total += chunk.apply_ops_tiled(ops, tiles) if tiles is not None and len(ops[0]) >= 2 else chunk.apply_ops(ops)
                return total
            run()                                             # warm-up: plans into the cache
            chunk.sync()
            best, passes = 1e9, 0
            for _ in range(3):
                t0 = time.perf_counter()
                passes = run()
                chunk.sync()
                best = min(best, time.perf_counter() - t0)
            per_exec.append((passes, len(relayouts), best * 1e3))
        ms = [e[2] for e in per_exec]
        print(f"n={n} on {world} ranks ({k} local qubits), rank {rank}, {name}, {label}: passes / re-layouts per execution "
              f"{[(a, b) for a, b, _ in per_exec]}, compute ms per execution {[round(x, 1) for x in ms]} (mean {np.mean(ms):.1f}, "
              f"{np.mean(ms) / np.mean([e[0] for e in per_exec]):.2f} ms per pass), planning {plan_s:.1f} s, slots {(eng.layout_info or {}).get('slot_placement')}", flush=True)
chunk.close()
