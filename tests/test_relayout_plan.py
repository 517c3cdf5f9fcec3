"""ONE re-layout schedule, testable without GPUs: `qsim_comm_relayout_plan` is the pure function
behind the C ABI's `qsim_comm_relayout`; here every rank's plan is computed in one process and checked for
  * pairing: what rank a sends to b is what b expects from a, piece by piece, at the offsets both sides use -- the
    partner semantics of the reference's chunk groups (wenbo_engine/runner/single_node.py:222-245: partners differ
    exactly in the swapped chunk-index bits) with one chunk per rank;
  * agreement with the Python engine: `DistributedEngine.relayout` (runner/distributed.py) posts the same peers in
    the same order with the same piece sizes (its transfers are recorded by a DryBackend, runner/dry_run.py);
  * data: moving numpy slabs by the plan gives the state with local and rank bits swapped."""
import itertools

import numpy as np
import pytest

from quantum_simulations_amd.kernel.device import relayout_plan


def _plans(world, k, local_bits, global_bits, pieces):
    return [relayout_plan(r, world, k, local_bits, global_bits, pieces) for r in range(world)]


CASES = [(2, 22, [5], [0], 4), (4, 23, [3, 17], [1, 0], 4), (8, 24, [0, 9, 23], [2, 0, 1], 8), (8, 10, [4], [2], 4),
         (4, 6, [1, 5], [0, 1], 2), (8, 30, [28, 4, 11], [0, 1, 2], 4), (4, 30, [29], [1], 1)]


@pytest.mark.parametrize("world,k,local_bits,global_bits,pieces", CASES)
def test_every_send_meets_its_receive(world, k, local_bits, global_bits, pieces):
    plans = _plans(world, k, local_bits, global_bits, pieces)
    m = len(local_bits)
    slab = 1 << (k - m)
    assert len({p["pieces"] for p in plans}) == 1 and len({p["piece_amps"] for p in plans}) == 1
    n_p, part = plans[0]["pieces"], plans[0]["piece_amps"]
    assert n_p * part == slab and (n_p == 1 or part >= 1 << 20)
    for a, pa in enumerate(plans):
        own_a = sum(((a >> g) & 1) << i for i, g in enumerate(global_bits))
        assert pa["own_pattern"] == own_a and len(pa["peers"]) == (1 << m) - 1
        for peer, off in zip(pa["peers"], pa["slab_offsets"]):
            d = off // slab
            assert off == d * slab and d != own_a
            # the peer differs from a exactly in the swapped rank bits, and its pattern is d
            assert (peer ^ a) & ~sum(1 << g for g in global_bits) == 0
            assert sum(((peer >> g) & 1) << i for i, g in enumerate(global_bits)) == d
            # ... and it lists a at the offset of a's pattern: b's receive from a lands where a's own pattern says
            pb = plans[peer]
            j = pb["peers"].index(a)
            assert pb["slab_offsets"][j] == own_a * slab


@pytest.mark.parametrize("world,k,local_bits,global_bits,pieces", [c for c in CASES if c[1] <= 12])
def test_moving_slabs_by_the_plan_swaps_local_and_rank_bits(world, k, local_bits, global_bits, pieces):
    rng = np.random.default_rng(3)
    full = rng.standard_normal(world << k) + 1j * rng.standard_normal(world << k)
    shards = [full[r << k:(r + 1) << k].copy() for r in range(world)]
    m = len(local_bits)
    slab = 1 << (k - m)
    idx = np.arange(1 << k)
    pat = sum(((idx >> b) & 1) << i for i, b in enumerate(local_bits))
    plans = _plans(world, k, local_bits, global_bits, pieces)
    send = [np.concatenate([s[pat == d] for d in range(1 << m)]) for s in shards]          # qsim_pack_all
    recv = [np.zeros(1 << k, dtype=complex) for _ in range(world)]
    for a, pa in enumerate(plans):
        for peer, off in zip(pa["peers"], pa["slab_offsets"]):
            for s in range(pa["pieces"]):
                lo = off + s * pa["piece_amps"]
                recv[a][lo:lo + pa["piece_amps"]] = send[peer][lo - off + plans[peer]["slab_offsets"][plans[peer]["peers"].index(a)]:][:pa["piece_amps"]]
    out = []
    for a, pa in enumerate(plans):
        s = shards[a].copy()
        for d in range(1 << m):
            if d != pa["own_pattern"]:
                s[pat == d] = recv[a][d * slab:(d + 1) * slab]                                # qsim_unpack_all
        out.append(s)
    got = np.concatenate(out)
    src = np.arange(world << k)
    for lb, gb in zip(local_bits, global_bits):
        g = k + gb
        x = ((src >> g) ^ (src >> lb)) & 1
        src = src ^ (x << g) ^ (x << lb)
    np.testing.assert_array_equal(got, full[src])


@pytest.mark.parametrize("world,k,pairs", [(2, 26, [[7, 26]]), (4, 25, [[3, 26], [20, 25]]), (8, 24, [[0, 24], [9, 26], [23, 25]])])
def test_python_engine_posts_the_planned_transfers(world, k, pairs):
    """runner/distributed.py and the C ABI compute the schedule separately: same peers, same order, same sizes."""
    from quantum_simulations_amd.runner.distributed import DistributedEngine, DryBackend
    n = k + world.bit_length() - 1
    for rank in range(world):
        for fuse in (False, True):
            # unfused: the pack / transfer / unpack pipeline in pieces; fused (slab bits above the line bits; round 4): the
            # slab-storing pass is cut into the same pieces and each piece's group is posted as soon as it is stored
            eng = DistributedEngine(n, world, rank, backend=DryBackend(k), init_process_group=False, fuse_relayout=fuse)
            eng.relayout(pairs)
            plan = relayout_plan(rank, world, k, [min(p) for p in pairs], [max(p) - k for p in pairs], eng.relayout_pieces)
            assert eng.trace_posts == plan["pieces"] >= 2
            posted = [(peer, sent // 16) for _, peer, sent, _ in eng.trace]
            want = [(peer, plan["piece_amps"]) for _ in range(plan["pieces"]) for peer in plan["peers"]]
            assert posted == want, (rank, fuse, posted[:8], want[:8])


def test_bad_arguments():
    with pytest.raises(ValueError):
        relayout_plan(0, 3, 10, [1], [0])
    with pytest.raises(ValueError):
        relayout_plan(0, 4, 10, [1, 1], [0, 1])
    with pytest.raises(ValueError):
        relayout_plan(0, 4, 10, [1], [2])
    with pytest.raises(NotImplementedError, match="non-local"):
        relayout_plan(0, 4, 10, [10], [0])
    with pytest.raises(ValueError):
        relayout_plan(0, 4, 10, [1], [0], 3)


def test_split_piece_rule_is_the_same_in_python_and_in_the_library():
    """The pieces a fused re-layout is cut into depend only on (local qubits, slab bits, pieces asked for): the library's
    rule (qsim_split_piece_count, a pure function: no device) and its restatement for the dry-run / CPU backends
    (runner/distributed.split_pieces) agree, pieces tile the slab, and the real floor keeps them >= 2^20 amplitudes."""
    from quantum_simulations_amd.kernel.device import split_piece_count
    from quantum_simulations_amd.runner.distributed import split_pieces
    for k in range(4, 34):
        for m in (1, 2, 3):
            if m >= k:
                continue
            for parts in (1, 2, 4, 8, -1, -2, -4, -8):
                pieces = split_pieces(k, m, parts)
                assert len(pieces) == split_piece_count(k, m, parts), (k, m, parts)
                assert len(pieces) <= abs(parts) and len(pieces) & (len(pieces) - 1) == 0
                assert [off for off, _ in pieces] == [j * pieces[0][1] for j in range(len(pieces))]
                assert sum(cnt for _, cnt in pieces) == 1 << (k - m)
                if parts > 1 and len(pieces) > 1:
                    assert pieces[0][1] >= 1 << 20
    assert len(split_pieces(30, 3, 4)) == 4 and len(split_pieces(30, 1, 8)) == 8 and len(split_pieces(22, 2, 4)) == 1
