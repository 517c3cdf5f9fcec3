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


def _worker(rank, world, port, n, errors):
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
            for name, cd in _circuits(n).items():
                want = orc.simulate(validate_circuit_dict(cd))
                eng.init_zero_state()
                eng.execute(eng.plan(cd))
                err = float(np.max(np.abs(eng.state_vector() - want)))
                assert err < 1e-10, f"{name} staging={staging} world={world}: {err}"
                assert abs(eng.norm2() - 1.0) < 1e-12
            eng.backend.close()
        eng.close()
    except Exception:
        errors.put((rank, traceback.format_exc()))
        raise


@pytest.mark.parametrize("world,n", [(2, 10), (4, 11)])
def test_ranks_sharing_one_gpu(world, n):
    ctx = mp.get_context("spawn")
    errors = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, errors)) for r in range(world)]
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
