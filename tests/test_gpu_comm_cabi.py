"""Multi-GPU reach of the C ABI (include/qsim_hip.h, qsim_comm_*): RCCL inside libqsim_hip.so, driven
WITHOUT torch.distributed.  One GPU is visible here, so this is the world-1 plumbing: unique id,
communicator, a grouped send / receive with itself, the partner-chunk butterfly fed by the exchange, and
the argument checks of the re-layout.  (RCCL refuses several ranks on one device; the multi-rank schedule
is covered by the gloo tests and the dry run.)"""
import numpy as np
import pytest

from oracle import dense_oracle as orc
from quantum_simulations_amd.kernel import gates as gt
from quantum_simulations_amd.kernel.device import Comm, DeviceChunk

pytestmark = pytest.mark.gpu


def _rand(k, seed):
    rng = np.random.default_rng(seed)
    v = rng.standard_normal(1 << k) + 1j * rng.standard_normal(1 << k)
    return v / np.linalg.norm(v)


def test_world1_exchange_and_remote_pair_through_the_c_abi():
    uid = Comm.unique_id()
    assert len(uid) == 128 and any(uid)
    comm = Comm(0, 0, 1, uid)
    k = 12
    a = _rand(k, 1)
    state, buf = DeviceChunk.from_numpy(a), DeviceChunk.zero_state(k, set_amp0=False)
    # two slices to "peer" 0 (itself) in one RCCL group
    half = 1 << (k - 1)
    comm.exchange([0, 0], state, [0, half], buf, [half, 0], half)
    state.sync()
    np.testing.assert_array_equal(buf.download(), np.concatenate([a[half:], a[:half]]))
    # partner-chunk butterfly: the "partner" shard arrives through RCCL, then cpu_nonlocal.apply_1q_pair
    U = gt.RY(0.7) @ gt.H()
    comm.apply_1q_pair_remote(state, buf, 0, 0, U)
    state.sync()
    c0, c1 = a.copy(), a.copy()
    orc.apply_1q_pair(c0, c1, U)
    np.testing.assert_allclose(state.download(), c0, rtol=0, atol=1e-14)
    state.upload(a)
    comm.apply_1q_pair_remote(state, buf, 0, 1, U)
    state.sync()
    np.testing.assert_allclose(state.download(), c1, rtol=0, atol=1e-14)
    # 2q with the global qubit as MSB (qb local) / LSB (qa local)
    V = np.linalg.qr(np.random.default_rng(5).standard_normal((4, 4)) + 1j * np.random.default_rng(6).standard_normal((4, 4)))[0]
    for fn, ref, q in ((comm.apply_2q_pair_qb_local_remote, orc.apply_2q_pair_qb_local, 3),
                       (comm.apply_2q_pair_qa_local_remote, orc.apply_2q_pair_qa_local, 7)):
        state.upload(a)
        fn(state, buf, 0, 0, q, V)
        state.sync()
        c0, c1 = a.copy(), a.copy()
        ref(c0, c1, q, V)
        np.testing.assert_allclose(state.download(), c0, rtol=0, atol=1e-14)
    # argument checks of the re-layout (a world of 1 has no rank bits)
    b0, b1 = DeviceChunk.zero_state(k), DeviceChunk.zero_state(k)
    with pytest.raises(ValueError):
        comm.relayout(state, b0, b1, [3], [0])
    with pytest.raises(NotImplementedError, match="non-local"):
        comm.relayout(state, b0, b1, [k], [0])
    with pytest.raises(ValueError):
        comm.exchange([1], state, [0], buf, [0], 4)          # peer out of range
    with pytest.raises(ValueError):
        comm.exchange([0], state, [half], buf, [0], half + 1)  # slice outside the chunk
    comm.close()
    for c in (state, buf, b0, b1):
        c.close()
