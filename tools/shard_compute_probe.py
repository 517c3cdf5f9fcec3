#!/usr/bin/env python3
"""One GPU: what the partition's initial layout is worth in per-GPU COMPUTE time.  Rank 0's staged schedule of a circuit on
N ranks is planned under the identity and under the layout `DistributedEngine.choose_initial_layout` picks; the op lists
rank 0 would hand to the library (recorded by the engine's planning twin) run on ONE shard-sized chunk (same kernels, same plans as on the node; the exchanges are skipped -- the
data is meaningless, the timing is not; the slab stores of the fused re-layouts are not part of it) and are timed.
    python tools/shard_compute_probe.py [N_QUBITS N_RANKS]"""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from quantum_simulations_amd import circuits as gen  # noqa: E402
from quantum_simulations_amd.circuit.io import validate_circuit_dict  # noqa: E402
from quantum_simulations_amd.kernel.device import DeviceChunk, pack_ops  # noqa: E402
from quantum_simulations_amd.runner.distributed import DistributedEngine, DryBackend  # noqa: E402

n, world = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (32, 4)
p = world.bit_length() - 1
k = n - p
eng = DistributedEngine(n, world, 0, backend=DryBackend(k), init_process_group=False, layout="search")
chunk = DeviceChunk.empty(k)
chunk.init_random(3)
for name, cd in (("random 1q+CX depth 40", gen.random_1q_cx_circuit(n, depth=40)), ("Clifford+T depth 60", gen.random_clifford_t_circuit(n, depth=60)),
                 ("GHZ+QFT", gen.generate_ghz_qft(n))):
    cd = validate_circuit_dict(cd)
    eng.init_zero_state()
    chosen = eng.choose_initial_layout(cd)
    info = eng.layout_info
    row = []
    for label, l2p in (("identity", list(range(n))), ("chosen", chosen)):
        eng._candidate_cost(cd, l2p)                        # (makes the shadow engine)
        eng._shadow.backend.record = []
        eng._candidate_cost(cd, l2p)                        # the op lists exactly as rank 0 would hand them to the library:
        lists = [pack_ops(ops) for ops in eng._shadow.backend.record]      # deferred batches merged, rank-bit phases and
        eng._shadow.backend.record = None                   # conditional gates of rank 0 included
        for ops in lists:                                   # warm-up: plans into the cache
            chunk.apply_ops(ops)
        chunk.sync()
        best, passes = 1e9, 0
        for _ in range(3):
            t0 = time.perf_counter()
            passes = sum(chunk.apply_ops(ops) for ops in lists)
            chunk.sync()
            best = min(best, time.perf_counter() - t0)
        row.append((label, passes, best * 1e3))
    (_, p0, t0_), (_, p1, t1_) = row
    print(f"n={n} on {world} ranks ({k} local qubits), {name}: identity {p0} passes {t0_:.1f} ms  ->  chosen {p1} passes {t1_:.1f} ms "
          f"({(t0_ / t1_ - 1) * 100:+.1f} % compute rate); model {info['identity']['cost_max_over_ranks']} -> {info['chosen']['cost_max_over_ranks']} pass units, "
          f"re-layouts {info['identity']['relayouts']} -> {info['chosen']['relayouts']}", flush=True)
chunk.close()
