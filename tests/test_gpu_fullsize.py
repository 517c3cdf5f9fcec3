"""Amplitude-level checks at BASELINE sizes.

Config 3 (30 qubits, one gate per launch, every target index): the H.H = I round trip of
test_gpu_kernels cannot see an involutory mistake (H on the wrong bit, twice).  Here sampled amplitude
pairs (i, i ^ 2^q) are downloaded BEFORE each gate, the butterfly of cpu_scalar.apply_1q
(wenbo_engine/kernel/cpu_scalar.py:21-32) is evaluated on the host, and the device values AFTER one
application must match at 1e-12 -- for H on every target, and for the secondary rows T(q), CNOT(q, q+1),
CNOT(0, q) (cpu_scalar.apply_2q semantics, :35-47).

Config 4 (Clifford+T, depth 60) at n = 26: every amplitude against the C oracle, (a) 4 ranks sharing the
GPU with the DEFAULT re-layout pipeline (relayout_pieces = 4, pieces of 2^20 amplitudes: the slabs of a
24-qubit shard really split), staged and unstaged (swap-and-stay), (b) the chunked single-GPU runner
single_node.run(chunk_size = 2^24, use_staging = True).
"""
import os
import sys
import traceback
from pathlib import Path

import numpy as np
import pytest
import torch.multiprocessing as mp

from oracle import c_oracle
from oracle import dense_oracle as orc
from tests.test_distributed_gloo import _free_port
from tests.test_gpu_kernels import hip  # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _sample_runs(n: int, bits, rng, runs: int = 48, run_len: int = 64):
    """Index runs [b, b + run_len) for every combination of `bits` around random bases with those bits clear
    (run_len amplitudes stay inside one combination when every bit is >= log2(run_len); lower bits are
    covered because a run of 64 contains every value of index bits 0..5)."""
    out = []
    for _ in range(runs):
        base = int(rng.integers(0, 1 << n)) & ~(run_len - 1)
        for b in bits:
            if b >= 6:
                base &= ~(1 << b)
        bases = [base]
        for b in bits:
            if b >= 6:
                bases += [x | (1 << b) for x in bases]
        out.append(sorted(set(bases)))
    return out


def _download_sample(dev, groups, run_len=64):
    return {b: dev.download(b, run_len) for g in groups for b in g}


def _host_apply(sample: dict, qubits, U, run_len=64):
    """Apply the gate to the sampled amplitudes on the host: every sampled index has its partners sampled."""
    out = {b: v.copy() for b, v in sample.items()}
    def get(i):
        b = i & ~(run_len - 1)
        return sample[b][i - b]
    def put(i, v):
        b = i & ~(run_len - 1)
        out[b][i - b] = v
    done = set()
    for b0 in sample:
        for j in range(run_len):
            i = b0 + j
            base = i
            for q in qubits:
                base &= ~(1 << q)
            if base in done:
                continue
            done.add(base)
            if len(qubits) == 1:
                idx = [base, base | (1 << qubits[0])]
            else:
                qa, qb = qubits
                idx = [base, base | (1 << qb), base | (1 << qa), base | (1 << qa) | (1 << qb)]
            vec = np.array([get(i2) for i2 in idx])
            res = U @ vec
            for i2, r in zip(idx, res):
                put(i2, r)
    return out


def test_config3_every_target_amplitudes_30q(hip):
    n = 30
    dev = hip.DeviceChunk.empty(n)
    dev.init_random(30)
    rng = np.random.default_rng(3030)
    H, T, CX = orc.gate_matrix("H"), orc.gate_matrix("T"), orc.gate_matrix("CNOT")
    cases = [([q], H, "H") for q in range(n)]
    cases += [([q], T, "T") for q in (0, 1, 2, 3, 7, 12, 19, 20, 25, 29)]
    cases += [([q, q + 1], CX, "CNOT(q,q+1)") for q in (0, 1, 2, 5, 6, 11, 19, 23, 28)]
    cases += [([0, q], CX, "CNOT(0,q)") for q in (1, 2, 3, 9, 20, 29)]
    worst = 0.0
    for qubits, U, label in cases:
        groups = _sample_runs(n, qubits, rng)
        before = _download_sample(dev, groups)
        want = _host_apply(before, qubits, U)
        if len(qubits) == 1:
            dev.apply_1q(qubits[0], U)
        else:
            dev.apply_2q(qubits[0], qubits[1], U)
        after = _download_sample(dev, groups)
        err = max(float(np.max(np.abs(after[b] - want[b]))) for b in want)
        changed = max(float(np.max(np.abs(after[b] - before[b]))) for b in want)
        assert err < 1e-12, f"{label} on {qubits}: max |device - host butterfly| = {err}"
        assert changed > 1e-9, f"{label} on {qubits}: the gate changed nothing in the sample"
        worst = max(worst, err)
    assert abs(dev.norm2() - 1.0) < 1e-11
    dev.close()


def test_dense_blocks_on_the_matrix_cores_at_30_qubits(hip):
    """qsim_apply_fused_k at full size (k_dense_mfma2: 16 GiB, the streaming instantiation, real offsets beyond 2^32): dense
    random unitaries on 3 to 6 qubits -- low, high, mixed and line bits, in the caller's (unsorted) order -- against the
    same contraction on the host for sampled blocks (every sampled amplitude has its 2^k - 1 partners sampled)."""
    n = 30
    dev = hip.DeviceChunk.empty(n)
    dev.init_random(31)
    rng = np.random.default_rng(3131)
    run_len = 64
    for qubits in ([29, 3, 17], [28, 29, 27], [5, 4, 3, 6], [29, 26, 27, 28], [12, 29, 4, 20], [1, 25, 0, 9], [2, 0, 1],
                   [7, 29, 3, 18, 11], [2, 27, 0, 1, 14], [29, 25, 28, 26, 27], [21, 4, 9, 29, 13, 6], [1, 0, 2, 3, 5, 4],
                   [29, 24, 26, 28, 25, 27]):
        k = len(qubits)
        M = np.linalg.qr(rng.standard_normal((1 << k, 1 << k)) + 1j * rng.standard_normal((1 << k, 1 << k)))[0]
        groups = _sample_runs(n, qubits, rng, runs=24)
        before = _download_sample(dev, groups)
        want = {b: v.copy() for b, v in before.items()}
        done = set()
        for b0 in before:
            for j in range(run_len):
                base = b0 + j
                for q in qubits:
                    base &= ~(1 << q)
                if base in done:
                    continue
                done.add(base)
                idx = [base | sum(((pat >> i) & 1) << q for i, q in enumerate(qubits)) for pat in range(1 << k)]
                vec = np.array([before[i & ~(run_len - 1)][i & (run_len - 1)] for i in idx])
                for i, r in zip(idx, M @ vec):
                    want[i & ~(run_len - 1)][i & (run_len - 1)] = r
        dev.apply_fused_k(qubits, M)
        after = _download_sample(dev, groups)
        err = max(float(np.max(np.abs(after[b] - want[b]))) for b in want)
        changed = max(float(np.max(np.abs(after[b] - before[b]))) for b in want)
        assert err < 1e-12, f"dense block on {qubits}: max |device - host| = {err}"
        assert changed > 1e-9, qubits
    assert abs(dev.norm2() - 1.0) < 1e-11
    dev.close()


def test_slab_layouts_at_30_qubits_64_bit_offsets(hip):
    """qsim_apply_ops_io at full shard size (30 local qubits, the multi-GPU configuration): slab bits at the top of the
    index push tile bits to physical positions >= 28 (the 64-bit-offset instantiations of k_tile) on both sides.
    A circuit goes out through the slab layout of one bit set and its inverse comes back in through it: sampled
    amplitudes of the result must equal the initial state (a wrong tile position anywhere breaks the identity)."""
    from quantum_simulations_amd.circuits import random_1q_cx_circuit
    from quantum_simulations_amd.kernel import gates as gt
    n = 30
    state, send, recv = hip.DeviceChunk.empty(n), hip.DeviceChunk.empty(n), hip.DeviceChunk.empty(n)
    plain, packed = hip.DeviceChunk.empty(n), hip.DeviceChunk.empty(n)
    state.init_random(77)
    rng = np.random.default_rng(5)
    windows = [int(o) for o in rng.integers(0, (1 << n) - (1 << 12), size=24)] + [0, (1 << n) - (1 << 12)]
    before = {o: state.download(o, 1 << 12) for o in windows}
    cd = random_1q_cx_circuit(n, depth=6, seed=30)
    ops = [(g["qubits"], gt.gate_matrix(g["gate"], g.get("params", {}))) for g in cd["gates"]]
    inv = [(qs, U.conj().T) for qs, U in reversed(ops)]
    for bits, own in (([29, 28], 2), ([28, 12, 29], 5), ([5], -1)):
        m = len(bits)
        slab = (1 << n) >> m
        windows = [o for o in windows if o // slab == (o + (1 << 12) - 1) // slab]      # (windows inside one slab)
        plain.copy_from(state)
        plain.apply_ops(ops)
        plain.pack_all(bits, packed, -1)                      # the independent slab kernel: what the fused store must equal
        recv.init_zero(False)
        p1 = state.apply_ops_io(ops, dst=(send, bits, recv if own >= 0 else None, own))
        for o in windows:
            d = o // slab
            got = (recv if d == own else send).download(o, 1 << 12)
            np.testing.assert_allclose(got, packed.download(o, 1 << 12), rtol=0, atol=1e-13, err_msg=f"{bits} slab {d} at {o}")
        # "exchange" with itself: every slab but the own one moves from the send to the receive buffer
        for d in range(1 << m):
            if d != own:
                recv.view(d * slab, n - m).copy_from(send.view(d * slab, n - m))
        p2 = state.apply_ops_io(inv, src=(recv, bits))
        assert p1 >= 1 and p2 >= 1
        for o in windows:
            err = float(np.max(np.abs(state.download(o, 1 << 12) - before[o])))
            assert err < 1e-11, (bits, own, o, err)
    assert abs(state.norm2() - 1.0) < 1e-10
    for c in (state, send, recv, plain, packed):
        c.close()


N4 = 26


def _config4_circuit():
    from quantum_simulations_amd.circuits import random_clifford_t_circuit
    return random_clifford_t_circuit(N4, depth=60)


def _oracle_state(cd):
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    c_oracle.set_threads(max(1, min(16, cores)))
    from quantum_simulations_amd.circuit.io import validate_circuit_dict
    return c_oracle.simulate(validate_circuit_dict(cd))


def _worker(rank, world, port, errors, results):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        sys.path.insert(0, str(ROOT))
        from quantum_simulations_amd.circuit.io import validate_circuit_dict
        from quantum_simulations_amd.runner.distributed import DistributedEngine, HipShardBackend
        cd = _config4_circuit()
        want = _oracle_state(cd) if rank == 0 else None
        p = world.bit_length() - 1
        # the default pipeline (layout search on: "auto" at 24 local qubits) staged and unstaged, then the four combinations
        # of staging x fused re-layouts in ONE layout (the identity), whose pass counts are compared below
        for staging, fuse, layout in ((True, True, "auto"), (False, True, "auto"), (True, True, "identity"), (False, True, "identity"),
                                      (True, False, "identity"), (False, False, "identity")):
            eng = DistributedEngine(N4, world, rank, backend=HipShardBackend(N4 - p, 0), staging=staging, fuse_relayout=fuse, layout=layout)
            assert eng.relayout_pieces == 4 and eng.min_piece_qubits == 20          # the defaults
            assert eng._relayout_pieces(N4 - p - 2) == 4                             # slabs really split
            eng.init_zero_state()
            plan = eng.plan(cd)
            eng.execute(plan)
            # the layout search prices candidates on a planning twin: what the twin counts for a layout is
            # what this rank's library really does -- tile passes, fused and unfused re-layout ends, swap-and-stay moves
            twin = eng._candidate_cost(validate_circuit_dict(cd), plan.start_mappings[0])
            assert twin[1] == eng.last_passes, (rank, staging, fuse, layout, twin, eng.last_passes)
            if layout == "auto":
                info = eng.layout_info
                assert info is not None and info["chosen"]["passes_this_rank"] == eng.last_passes, (rank, info, eng.last_passes)
                assert info["chosen"]["cost_max_over_ranks"] <= info["identity"]["cost_max_over_ranks"]
            else:
                assert eng.layout_info is None and plan.start_mappings[0] == list(range(N4))
            stats = eng.comm_stats()
            norm2 = eng.norm2()
            got = eng.state_vector()                                                # gathered, logical order
            if rank == 0:
                err = float(np.max(np.abs(got - want)))
                if err > 1e-10:        # where: index range and the index bits the wrong amplitudes share / differ in
                    bad = np.flatnonzero(np.abs(got - want) > 1e-10)
                    print(f"[staging={staging}] {bad.size} wrong amplitudes, first {bad[:8].tolist()}, last {int(bad[-1])}, "
                          f"AND {int(np.bitwise_and.reduce(bad)):#x} OR {int(np.bitwise_or.reduce(bad)):#x}", file=sys.stderr, flush=True)
                results.put((staging, fuse, layout, err, norm2, stats["exchanges"], stats["bytes_sent_per_rank"], eng.last_passes))
            if fuse and layout == "auto":
                # the amplitude check that needs no gather: per-shard fingerprints in the staged /
                # moved layout against the same index sets of a ONE-device run of the circuit (the product hook: a
                # 26-qubit SingleGpuEngine on rank 0's device), then the same after two slabs traded places
                got_fp, sel = eng.fingerprints(5), eng.shard_selectors()
                (diff,) = eng.check_against_single_device(cd, [(got_fp, sel)], seed=5)
                assert diff < 1e-10, (staging, diff)
                if rank == 0:       # ... and against the C oracle's state through the numpy restatement of the weights
                    from tests.cpu_shard_backend import fingerprint_np
                    for r, (m, v) in enumerate(sel):
                        assert abs(fingerprint_np(want, N4, 0, None, 5, m, v) - got_fp[r]) < 1e-10, (staging, r)
                t = eng.backend.tensor("state")
                half = t.numel() // 2
                lo = t[:half].clone()
                t[:half].copy_(t[half:])
                t[half:].copy_(lo)
                del lo
                assert abs(eng.norm2() - 1.0) < 1e-10                                # the norm does not see it
                (bad_fp,) = eng.check_against_single_device(cd, [(eng.fingerprints(5), sel)], seed=5)
                assert bad_fp > 1e-4, bad_fp
            del got
            eng.backend.close()
        eng.close()
    except Exception:
        errors.put((rank, traceback.format_exc()))
        raise


def test_config4_clifford_t_26q_four_ranks_default_pipeline():
    world = 4
    ctx = mp.get_context("spawn")
    errors, results = ctx.SimpleQueue(), ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, errors, results)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(900)
    msgs = []
    while not errors.empty():
        msgs.append(errors.get())
    for p in procs:
        if p.is_alive():
            p.terminate()
            msgs.append((-1, "timeout"))
    assert not msgs and all(p.exitcode == 0 for p in procs), "\n".join(f"[rank {r}] {m}" for r, m in msgs)
    seen, passes = {}, {}
    while not results.empty():
        staging, fuse, layout, err, norm2, exchanges, sent, n_passes = results.get()
        if layout == "identity":
            passes[(staging, fuse)] = (n_passes, exchanges)
        else:
            seen[staging] = (err, exchanges, sent)
        assert err < 1e-10, f"staging={staging} fuse={fuse} layout={layout}: max |amp - C oracle| = {err}"
        assert abs(norm2 - 1.0) < 1e-10
    assert set(seen) == {True, False} and len(passes) == 4
    assert 0 < seen[True][1] <= seen[False][1]                  # staging needs no more exchanges than swap-and-stay
    shard = 16 << (N4 - 2)
    assert seen[False][2] <= seen[False][1] * shard * 3 // 4    # swap-and-stay: at most 3/4 of a shard per move
    # re-layouts fused into the neighbouring tile passes: same exchanges, same bytes, and the
    # HBM passes of rank 0 drop by the pack + unpack of (nearly) every re-layout -- a pack / unpack survives only
    # where no local pass is adjacent or a slab bit is a tile bit of the last pass
    for staging in (True, False):
        (fused, ex_f), (plain, ex_p) = passes[(staging, True)], passes[(staging, False)]
        print(f"staging={staging}: {plain} -> {fused} HBM passes on rank 0 for {ex_p} re-layouts")
        assert ex_f == ex_p
        assert plain - fused >= 2 * ex_p - 3, (staging, plain, fused, ex_p)


def test_config4_clifford_t_26q_chunked_single_gpu_runner(hip):
    from quantum_simulations_amd.runner import single_node
    cd = _config4_circuit()
    want = _oracle_state(cd)
    buf = single_node.run(cd, None, chunk_size=1 << 24, use_staging=True)
    got = single_node.collect_state(buf, apply_permutation=False)
    if buf.log_to_phys is not None:
        from quantum_simulations_amd.circuit.staging import permute_state
        got = permute_state(got, buf.log_to_phys)
    err = float(np.max(np.abs(got - want)))
    buf.close()
    assert err < 1e-10, err
