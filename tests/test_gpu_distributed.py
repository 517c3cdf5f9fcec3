"""Multi-rank rehearsal on ONE MI355X: 2 and 4 ranks share cuda:0, each holding its shard in
HBM and running the real HIP kernels; the exchange goes over gloo (host-staged) because RCCL
refuses several ranks on one device.  Same schedule code as the 8-GPU run."""
import os
import sys
import traceback
from pathlib import Path

import numpy as np
import pytest
import torch.multiprocessing as mp

from tests.test_distributed_gloo import _circuits, _free_port

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _worker(rank, world, port, n, errors, moves):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                          RANK=str(rank), WORLD_SIZE=str(world))
        sys.path.insert(0, str(ROOT))
        from oracle import dense_oracle as orc
        from quantum_simulations_amd.circuit.io import validate_circuit_dict
        from quantum_simulations_amd.runner.distributed import DistributedEngine, HipShardBackend
        p = world.bit_length() - 1
        for staging in (True, False):
            eng = DistributedEngine(n, world, rank, backend=HipShardBackend(n - p, 0), staging=staging,
                                    relayout_pieces=4 if world == 2 else 2, min_piece_qubits=1)
            circuits = dict(_circuits(n))
            # dense gates on the global qubits in turn with ONE local gate between them: unstaged, every one is a swap-and-stay
            # re-layout whose queued op list (the gate itself, now local) reads the receive buffer and stores the next slabs
            # in one pass -- the own slab goes into "state" and the buffers trade names (three buffers per rank)
            circuits["pingpong"] = {"number_of_qubits": n, "gates": [g for r in range(3) for g in (
                {"qubits": [n - 1], "gate": "H"}, {"qubits": [n - 2], "gate": "RY", "params": {"theta": 0.3 + r}},
                {"qubits": [n - 1, 4], "gate": "CNOT"})]}
            for name, cd in circuits.items():
                want = orc.simulate(validate_circuit_dict(cd))
                eng.init_zero_state()
                eng.execute(eng.plan(cd))
                err = float(np.max(np.abs(eng.state_vector() - want)))
                assert err < 1e-10, f"{name} staging={staging} world={world}: {err}"
                assert abs(eng.norm2() - 1.0) < 1e-12
            moves.put((staging, eng.home_moves))
            eng.backend.close()
        eng.close()
    except Exception:
        errors.put((rank, traceback.format_exc()))
        raise


@pytest.mark.parametrize("world,n", [(2, 10), (4, 11), (4, 16)])
def test_ranks_sharing_one_gpu(world, n):
    ctx = mp.get_context("spawn")
    errors, moves = ctx.SimpleQueue(), ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, errors, moves)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
    msgs = []
    while not errors.empty():
        msgs.append(errors.get())
    for p in procs:
        if p.is_alive():
            p.terminate()
            msgs.append((-1, "timeout"))
    assert not msgs and all(p.exitcode == 0 for p in procs), "\n".join(f"[rank {r}] {m}" for r, m in msgs)
    total = 0
    while not moves.empty():
        total += moves.get()[1]
    assert n < 16 or total > 0, "no fused re-layout took the one-pass branch (own slab into the chunk, state / buf1 trade names)"
