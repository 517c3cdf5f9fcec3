#!/usr/bin/env python3
"""Randomised GPU-vs-oracle stress: fused and per-gate op lists of every gate kind on n = 1..20
qubits, chunked runs with random chunk sizes, staged/unstaged.  A script, not collected by pytest;
it lives under tests/ because only test code may use oracle/.
    python tests/stress_gpu.py [seconds] [seed]"""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from oracle import dense_oracle as orc  # noqa: E402
from quantum_simulations_amd.kernel.device import DeviceChunk  # noqa: E402
from quantum_simulations_amd.runner import single_node  # noqa: E402
from tests.test_gpu_kernels import _rand_state, _random_ops  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2026)
t0 = time.time()
cases = 0
worst = 0.0
three_in_chunk = three_in_source = 0
while time.time() - t0 < budget:
    n = int(rng.integers(1, 21))
    seed = int(rng.integers(1 << 30))
    ops = _random_ops(n, int(rng.integers(1, 120)), seed) if n > 1 else [([0], orc.gate_matrix("H"))] * 5
    psi0 = _rand_state(n, seed)
    want = psi0.copy()
    orc.apply_ops(want, ops)
    dev = DeviceChunk.from_numpy(psi0)
    for fused in (True, False):
        dev.upload(psi0)
        dev.apply_ops(ops, fused=fused)
        err = float(np.max(np.abs(dev.download() - want)))
        worst = max(worst, err)
        assert err < 1e-10, (n, seed, fused, err)
    if n >= 8:               # the caller names the tiles of the first passes (qsim_apply_ops_tiled): ANY masks -- useful,
        hints = []           # useless (no op fits: ignored), too wide, inside the line -- must leave the result alone
        for _ in range(int(rng.integers(1, 6))):
            width = int(rng.integers(0, 11))
            hints.append(sum(1 << int(b) for b in rng.choice(n, size=min(width, n), replace=False)))
        dev.upload(psi0)
        dev.apply_ops_tiled(ops, np.array(hints, dtype=np.uint64))
        err = float(np.max(np.abs(dev.download() - want)))
        worst = max(worst, err)
        assert err < 1e-10, ("tiled", n, seed, hints, err)
    if n >= 3:               # dense k-qubit blocks, k <= 6 (small chunks: one workgroup per block; else the matrix cores)
        k = int(rng.integers(3, min(6, n) + 1))
        qs = [int(q) for q in rng.choice(n, size=k, replace=False)]
        M = np.linalg.qr(rng.standard_normal((1 << k, 1 << k)) + 1j * rng.standard_normal((1 << k, 1 << k)))[0]
        wantk = psi0.copy()
        orc.apply_kq(wantk, qs, M)
        dev.upload(psi0)
        dev.apply_fused_k(qs, M)
        err = float(np.max(np.abs(dev.download() - wantk)))
        worst = max(worst, err)
        assert err < 1e-10, ("dense", n, seed, qs, err)
    if n >= 4:               # the same op list with a re-layout fused into its ends (qsim_apply_ops_io), random slab bits
        m = int(rng.integers(1, min(3, n - 1) + 1))
        bits_in = [int(b) for b in rng.choice(n, size=m, replace=False)]
        bits_out = [int(b) for b in rng.choice(n, size=m, replace=False)]
        own = int(rng.integers(-1, 1 << m))
        idx = np.arange(1 << n)

        def slabs(vec, bits):
            pat = sum(((idx >> b) & 1) << i for i, b in enumerate(bits))
            return [vec[pat == d] for d in range(1 << len(bits))]
        src, dst, keep = DeviceChunk.from_numpy(np.concatenate(slabs(psi0, bits_in))), DeviceChunk.empty(n), DeviceChunk.empty(n)
        dst.init_zero(False)
        keep.init_zero(False)
        dev.init_zero(False)
        split = int(rng.integers(0, 4))       # 0: one call stores the slabs; else the split form, pieces stored in random order
        three = own >= 0 and rng.random() < 0.5      # three buffers: the own slab goes into the consumed source buffer -- or,
        own_buf = src if three else keep             # when one pass does everything, into the chunk itself (own_slab_in_chunk)
        io_hints = None
        if rng.random() < 0.5:                       # ... with the tiles of the first passes named by the caller (any masks)
            io_hints = np.array([sum(1 << int(b) for b in rng.choice(n, size=min(int(rng.integers(0, 10)), n), replace=False))
                                 for _ in range(int(rng.integers(1, 5)))], dtype=np.uint64)
        dev.apply_ops_io(ops, src=(src, bits_in), dst=(dst, bits_out, own_buf if own >= 0 else None, own), parts=-(1 << split) if split else 0,
                         tiles=io_hints)
        if split:
            parts = dev.pending_parts()
            slab_amps = (1 << n) >> m
            seen = np.zeros(slab_amps, dtype=np.int32)
            for j in rng.permutation(len(parts)):
                dev.store_part(int(j))
                off, cnt = parts[int(j)]
                seen[off:off + cnt] += 1
            assert np.all(seen == 1), ("pieces do not tile the slab", n, seed, bits_out, parts)
        in_chunk = dev.own_slab_in_chunk()
        assert not in_chunk or three, ("own slab in the chunk without being asked", n, seed)
        three_in_chunk += in_chunk
        three_in_source += three and not in_chunk
        g0, g1 = dst.download(), (dev if in_chunk else own_buf).download()
        slab = (1 << n) >> m
        for d, w in enumerate(slabs(want, bits_out)):
            err = float(np.max(np.abs((g1 if d == own else g0)[d * slab:(d + 1) * slab] - w)))
            worst = max(worst, err)
            assert err < 1e-10, ("io", n, seed, bits_in, bits_out, own, d, three, in_chunk, err)
        for c in (src, dst, keep):
            c.close()
    dev.close()
    if n >= 2 and n <= 12:   # chunked runner with a random chunk size, from |0..0>
        names = ["H", "X", "T", "CNOT", "CZ", "SWAP", "CY"]
        gates = []
        for _ in range(int(rng.integers(1, 30))):
            g = names[int(rng.integers(len(names)))]
            if g in ("CNOT", "CZ", "SWAP", "CY"):
                a, b = (int(x) for x in rng.choice(n, size=2, replace=False))
                gates.append({"qubits": [a, b], "gate": g})
            else:
                gates.append({"qubits": [int(rng.integers(n))], "gate": g})
        cd = {"number_of_qubits": n, "gates": gates}
        ref = orc.simulate(cd)
        cs = 1 << int(rng.integers(0, n + 1))
        for kw in ({}, {"use_fusion": True}, {"use_staging": True, "staging_method": "belady"},
                   {"use_staging": True, "staging_method": "greedy"}):
            if kw.get("use_staging") and cs < 4 and kw["staging_method"] != "greedy":
                continue
            buf = single_node.run(cd, None, chunk_size=cs, **kw)
            got = single_node.collect_state(buf)
            if buf.log_to_phys:
                from quantum_simulations_amd.circuit.staging import permute_state
                got = permute_state(got, buf.log_to_phys)
            buf.close()
            err = float(np.max(np.abs(got - ref)))
            worst = max(worst, err)
            assert err < 1e-10, (n, cs, kw, err)
    cases += 1
    if cases % 50 == 0:
        print(f"... {cases} cases, {time.time() - t0:.0f} s, worst {worst:.2e}", flush=True)   # (a silent run is taken for hung)
# large states, every gate kind, against the C oracle: many waves per SIMD, full tiles, 32-bit and (n >= 29)
# 64-bit thread offsets are NOT reached here -- the 33-qubit tests of test_gpu_kernels cover those
from oracle import c_oracle  # noqa: E402
big_cases = 0
t1 = time.time()
while time.time() - t1 < budget * 0.5:
    n = int(rng.integers(21, 26))
    seed = int(rng.integers(1 << 30))
    ops = _random_ops(n, int(rng.integers(60, 200)), seed)
    psi0 = _rand_state(n, seed)
    want = psi0.copy()
    for qs, U in ops:
        (c_oracle.apply_1q if len(qs) == 1 else c_oracle.apply_2q)(want, *qs, np.ascontiguousarray(U, dtype=np.complex128))
    dev = DeviceChunk.from_numpy(psi0)
    dev.apply_ops(ops, fused=True)
    err = float(np.max(np.abs(dev.download() - want)))
    worst = max(worst, err)
    assert err < 1e-10, ("big", n, seed, err)
    # the same list with its slab-storing pass cut into pieces (qsim_ops_io::dst_parts), real 2^20 floor from 23 qubits on
    m = int(rng.integers(1, 4))
    bits = [int(b) for b in rng.choice(np.arange(3, n), size=m, replace=False)]
    own = int(rng.integers(0, 1 << m))
    dst, keep = DeviceChunk.empty(n), DeviceChunk.empty(n)
    dev.upload(psi0)
    dev.apply_ops_io(ops, dst=(dst, bits, keep, own), parts=4 if n >= 23 else -4)
    parts = dev.pending_parts()
    dev_launches = dev.last_split_launches
    for j in rng.permutation(len(parts)):
        dev.store_part(int(j))
    idx = np.arange(1 << n)
    pat = sum(((idx >> b) & 1) << i for i, b in enumerate(bits))
    slab = (1 << n) >> m
    g0, g1 = dst.download(), keep.download()
    for d in range(1 << m):
        e2 = float(np.max(np.abs((g1 if d == own else g0)[d * slab:(d + 1) * slab] - want[pat == d])))
        worst = max(worst, e2)
        assert e2 < 1e-10, ("big split", n, seed, bits, own, d, len(parts), e2)
    for c in (dev, dst, keep):
        c.close()
    big_cases += 1
    print(f"... large case {big_cases}: n = {n}, {len(ops)} ops, err {err:.2e}; slabs over {bits} in {len(parts)} piece(s), {dev_launches} partial launch(es)", flush=True)
print(f"stress ok: {big_cases} large cases (21-25 qubits, fused) in {time.time() - t1:.0f} s")
print(f"stress ok: {cases} random cases in {time.time() - t0:.0f} s, worst |diff| = {worst:.2e}; three-buffer re-layout ends: "
      f"{three_in_source} own slabs into the source buffer, {three_in_chunk} into the chunk")
