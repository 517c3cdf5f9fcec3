"""RCCL plumbing on the one GPU a test box has: a world-size-1 `nccl` process group must
initialise, reduce and gather CUDA float64 tensors wrapped by the engine's backend (the
multi-rank exchange itself is rehearsed in test_gpu_distributed.py over gloo and runs over
RCCL only on the 8-GPU node)."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_nccl_world1_collectives_on_wrapped_memory():
    import torch
    import torch.distributed as dist

    from quantum_simulations_amd.kernel import gates as gt
    from quantum_simulations_amd.runner.distributed import HipShardBackend
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        be = HipShardBackend(10, 0)
        be.init_zero(True)
        be.apply_ops([([q], gt.H()) for q in range(10)])
        t = be.tensor("state")
        total = (t * t).sum().reshape(1)
        dist.all_reduce(total)
        assert abs(float(total.item()) - 1.0) < 1e-12
        parts = [torch.empty_like(t)]
        dist.all_gather(parts, t)
        got = parts[0].cpu().numpy().view(np.complex128)
        np.testing.assert_allclose(got, 2.0 ** -5, atol=1e-14)
        # self send/recv through the batched P2P path the re-layout uses
        buf = be.tensor("buf1")
        ops = [dist.P2POp(dist.isend, t, 0), dist.P2POp(dist.irecv, buf, 0)]
        for w in dist.batch_isend_irecv(ops):
            w.wait()
        torch.cuda.synchronize()
        assert torch.equal(buf, t)
        dist.barrier()
        be.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("api", ["torch", "cabi"])
def test_world1_engine_rides_on_either_exchange(api):
    """DistributedEngine(exchange="torch" | "cabi") on the one GPU a test box has: a world of 1 over
    real RCCL -- process group, (cabi) the library's own communicator made from a broadcast unique id -- runs a circuit,
    reduces norm and fingerprints, and pushes a self-transfer through the engine's own _post / _finish (torch: batched
    P2P; cabi: qsim_comm_exchange_bg on the transfer stream + qsim_comm_join)."""
    import torch
    import torch.distributed as dist

    from oracle import dense_oracle as orc
    from quantum_simulations_amd import circuits as gen
    from quantum_simulations_amd.circuit.io import validate_circuit_dict
    from quantum_simulations_amd.runner.distributed import DistributedEngine
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    n = 14
    eng = DistributedEngine(n, 1, 0, 0, exchange=api)
    try:
        assert eng.exchange == "rccl" and eng.exchange_api == api and dist.get_backend() == "nccl"
        cd = gen.random_clifford_t_circuit(n, depth=20, seed=2)
        eng.init_zero_state()
        eng.execute(eng.plan(cd))
        want = orc.simulate(validate_circuit_dict(cd))
        np.testing.assert_allclose(eng.state_vector(), want, rtol=0, atol=1e-11)
        assert abs(eng.norm2() - 1.0) < 1e-12
        (fp,) = eng.fingerprints(3)
        from tests.cpu_shard_backend import fingerprint_np
        assert abs(fp - fingerprint_np(want, n, 0, None, 3)) < 1e-12
        # the engine's exchange path: two slices to "peer" 0 in one group, beside work queued later on the shard's stream
        b0, b1 = eng.backend.tensor("buf0"), eng.backend.tensor("buf1")
        b0.copy_(torch.arange(b0.numel(), dtype=torch.float64, device=b0.device))
        b1.zero_()
        half = b0.numel() // 2
        posted = eng._post("buf0", "buf1", [(0, 0, half // 2), (0, half, half // 2)])
        eng.backend.apply_ops([([3], orc.gate_matrix("H"))])          # (later work on the shard's stream)
        eng._finish(posted)
        torch.cuda.synchronize()
        assert torch.equal(b1[:half // 2], b0[:half // 2]) and torch.equal(b1[half:half + half // 2], b0[half:half + half // 2])
        assert not b1[half // 2:half].any()
        assert eng.xgmi_bytes_sent == half * 8
    finally:
        eng.close()
