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
