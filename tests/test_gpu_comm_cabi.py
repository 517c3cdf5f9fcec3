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


def test_relayout_pipeline_loopback_runs_the_piece_loop():
    """`qsim_comm_relayout`'s data path on ONE GPU: the schedule of rank `as_rank` in a world of
    2 / 4 / 8 (qsim_comm_relayout_plan) with every transfer looped back to this rank -- the same packs on the chunk's
    stream, events, transfers on the second stream in RCCL groups (piece by piece: 2^21-amplitude slabs are cut),
    and unpacks as a real re-layout.  The slabs come back unchanged, so the state must be bit-identical afterwards
    and the receive buffer must hold every slab but the own one at the planned offsets."""
    from quantum_simulations_amd.kernel.device import relayout_plan
    comm = Comm(0, 0, 1, Comm.unique_id())
    k = 23
    a = _rand(k, 11)
    state, b0, b1 = DeviceChunk.from_numpy(a), DeviceChunk.zero_state(k, set_amp0=False), DeviceChunk.zero_state(k, set_amp0=False)
    idx = np.arange(1 << k)
    for as_world, as_rank, lb, gb, pieces in ((2, 1, [7], [0], 4), (4, 2, [0, 19], [1, 0], 2), (8, 5, [22, 3, 11], [0, 2, 1], 4),
                                              (8, 0, [5], [1], 8)):
        plan = relayout_plan(as_rank, as_world, k, lb, gb, pieces)
        assert plan["pieces"] == min(pieces, 1 << (k - len(lb) - 20))          # pieces keep >= 2^20 amplitudes
        b1.init_zero(False)
        comm.relayout_loopback(state, b0, b1, lb, gb, pieces, as_rank, as_world)
        state.sync()
        np.testing.assert_array_equal(state.download(), a)
        got = b1.download()
        slab = 1 << (k - len(lb))
        pat = sum(((idx >> b) & 1) << i for i, b in enumerate(lb))
        for d in range(1 << len(lb)):
            want = a[pat == d] if d != plan["own_pattern"] else np.zeros(slab, dtype=complex)
            np.testing.assert_array_equal(got[d * slab:(d + 1) * slab], want, err_msg=f"world {as_world} slab {d}")
        assert sorted(o // slab for o in plan["slab_offsets"]) == [d for d in range(1 << len(lb)) if d != plan["own_pattern"]]
    with pytest.raises(ValueError):
        comm.relayout_loopback(state, b0, b1, [3], [3], 1, 0, 8)      # rank bit 3 does not exist in a world of 8
    comm.close()
    for c in (state, b0, b1):
        c.close()


def test_quad_remote_loopback_against_the_oracle():
    """qsim_apply_2q_quad_remote (cpu_nonlocal.py:61-67 with the four chunks on four ranks) in its one-GPU loopback form:
    every transfer comes back, so with all four chunks taking part chunk j of the group is quarter j of the shard and the
    result must equal oracle.apply_2q_quad on the four quarters -- whichever chunk this rank plays.  Dense 4x4 (all four chunks take part),
    a gate controlled by qa (chunks |10>, |11> only: a 2x2 across the pair, the other two ranks return at once), SWAP
    (chunks |01>, |10>), CZ (a phase on chunk |11> alone: no exchange at all), the identity."""
    comm = Comm(0, 0, 1, Comm.unique_id())
    k = 13
    a = _rand(k, 21)
    Q = 1 << (k - 2)
    rng = np.random.default_rng(3)
    dense = np.linalg.qr(rng.standard_normal((4, 4)) + 1j * rng.standard_normal((4, 4)))[0]
    cu = np.eye(4, dtype=complex)
    cu[2:, 2:] = gt.RY(0.9) @ gt.H()
    cases = {"dense": (dense, [0, 1, 2, 3]), "CNOT(qa,qb)": (gt.CNOT(), [2, 3]), "CU(qa,qb)": (cu, [2, 3]),
             "SWAP": (gt.SWAP(), [1, 2]), "CZ": (gt.CZ(), [3]), "identity": (np.eye(4, dtype=complex), [])}
    state, buf = DeviceChunk.from_numpy(a), DeviceChunk.zero_state(k, set_amp0=False)
    for name, (U, active) in cases.items():
        want = a.copy()
        if len(active) == 2:
            # two chunks take part: each works on every second quarter, so looped back the pair (chunk a, chunk b) is
            # (quarter 0, quarter 1) and (quarter 2, quarter 3): the 2x2 restriction of U on local qubit k - 2
            orc.apply_1q(want, k - 2, U[np.ix_(active, active)])
        elif len(active) == 1:
            want = a * U[active[0], active[0]]                                   # one chunk: its phase, on the shard that holds it
        else:
            orc.apply_2q_quad(*(want[j * Q:(j + 1) * Q] for j in range(4)), U)   # chunk j = quarter j
        for me in range(4):
            state.upload(a)
            comm.apply_2q_quad_remote(state, buf, [0, 0, 0, 0], me, U)
            state.sync()
            got = state.download()
            if me in active or name == "dense":
                np.testing.assert_allclose(got, want, rtol=0, atol=1e-13, err_msg=f"{name} as chunk {me}")
            else:
                np.testing.assert_array_equal(got, a, err_msg=f"{name}: chunk {me} takes no part")
    with pytest.raises(ValueError):
        comm.apply_2q_quad_remote(state, buf, [0, 0, 0, 1], 0, dense)        # rank out of range
    with pytest.raises(ValueError):
        comm.apply_2q_quad_remote(state, state, [0, 0, 0, 0], 0, dense)      # buffer = shard
    with pytest.raises(ValueError):
        comm.apply_2q_quad_remote(state, buf, [0, 0, 0, 0], 4, dense)
    comm.close()
    state.close()
    buf.close()


@pytest.mark.parametrize("k,pieces", [(13, -4), (16, -2), (23, 4)])
def test_relayout_fused_loopback_against_the_oracle(k, pieces):
    """qsim_comm_relayout_fused as rank `as_rank` of a world of 2 / 4 / 8 with every transfer looped back: the slabs this
    rank "receives" are its own, so the call must equal after(before(state)) of the oracle -- while running the real
    schedule: split last pass, per-piece RCCL groups on the transfer stream, events, first pass reading the receive
    buffer.  Pass counts: two fewer than pack + exchange + unpack around the same op lists."""
    from tests.test_gpu_kernels import _random_ops
    comm = Comm(0, 0, 1, Comm.unique_id())
    a = _rand(k, 31 + k)
    state, send, recv = DeviceChunk.from_numpy(a), DeviceChunk.zero_state(k, set_amp0=False), DeviceChunk.zero_state(k, set_amp0=False)
    rng = np.random.default_rng(50 + k)
    for trial, (as_world, as_rank, m) in enumerate(((2, 1, 1), (4, 2, 2), (8, 5, 3), (8, 0, 1), (4, 3, 2))):
        lb = [int(b) for b in rng.choice(np.arange(3 if trial != 3 else 0, k), size=m, replace=False)]
        gb = [int(g) for g in rng.permutation(as_world.bit_length() - 1)[:m]]
        before = _random_ops(k, 40, 100 * k + trial) if trial != 1 else []
        after = _random_ops(k, 40, 200 * k + trial) if trial != 2 else []
        want = a.copy()
        orc.apply_ops(want, before)
        orc.apply_ops(want, after)
        state.upload(a)
        plain = DeviceChunk.from_numpy(a)
        plain_passes = (plain.apply_ops(before) if before else 0) + (plain.apply_ops(after) if after else 0)
        plain.close()
        passes = comm.relayout_fused(state, send, recv, before, after, lb, gb, pieces, as_rank, as_world)
        state.sync()
        np.testing.assert_allclose(state.download(), want, rtol=0, atol=1e-11, err_msg=f"k={k} trial={trial} bits {lb}<->{gb}")
        assert passes <= plain_passes + 2, (trial, passes, plain_passes)
        if min(lb) >= 3 and before and after and k >= 16:
            assert passes <= plain_passes + 1, (trial, passes, plain_passes)     # at most one end not fused (slab bit = tile bit)
    with pytest.raises(ValueError):
        comm.relayout_fused(state, send, recv, [], [], [3], [3], 4, 0, 8)         # rank bit 3 does not exist in a world of 8
    with pytest.raises(ValueError):
        comm.relayout_fused(state, send, recv, [], [], [3], [0], 3, 0, 2)         # n_pieces
    comm.close()
    for c in (state, send, recv):
        c.close()
