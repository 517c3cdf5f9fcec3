"""Re-layout fused into the ends of an op list (qsim_apply_ops_io, SURVEY 8e / staging.py:136-152): the last fused
pass stores its tiles in the slab layout of qsim_pack_all (own slab into the receive buffer), the first one reads a
state that arrived in slab layout -- checked against oracle + numpy slabs for every combination the launcher
distinguishes: fusable, slab bit inside a 128-byte line (separate slab pass), chunk too small for tile passes, a slab
bit that is a tile bit of the last pass, no own slab.  Pass counts prove the fusion really happened."""
import numpy as np
import pytest

from oracle import dense_oracle as orc
from tests.test_gpu_kernels import _rand_state, _random_ops

pytestmark = pytest.mark.gpu


def _slabs(vec, bits):
    idx = np.arange(len(vec))
    pat = sum(((idx >> b) & 1) << i for i, b in enumerate(bits))
    return [vec[pat == d] for d in range(1 << len(bits))]


@pytest.mark.parametrize("k", [7, 11, 13, 16, 19])
def test_ops_with_fused_slab_output_and_input(k):
    from quantum_simulations_amd.kernel.device import DeviceChunk
    rng = np.random.default_rng(100 + k)
    state, buf0, buf1 = DeviceChunk.empty(k), DeviceChunk.empty(k), DeviceChunk.empty(k)
    n = 1 << k
    for trial in range(8):
        m = int(rng.integers(1, 4))
        lo = 0 if trial % 4 == 3 else 3                       # every fourth trial: a slab bit inside a 128-byte line
        bits = [int(b) for b in rng.choice(np.arange(lo, k), size=m, replace=False)]
        if trial % 4 == 3 and min(bits) >= 3:
            bits[0] = int(rng.integers(0, 3))
        own = int(rng.integers(-1, 1 << m)) if trial % 3 else int(rng.integers(0, 1 << m))
        ops = _random_ops(k, 50, 7000 + 31 * k + trial)
        if trial == 5:                                        # the last op targets a slab bit: it is a tile bit of the last pass
            ops.append(([bits[0]], orc.gate_matrix("H")))
        psi0 = _rand_state(k, 8000 + 17 * k + trial)
        want = psi0.copy()
        orc.apply_ops(want, ops)
        slab = n >> m
        plain = DeviceChunk.from_numpy(psi0)
        plain_passes = plain.apply_ops(ops)
        plain.close()
        fusable = k >= 8 and min(bits) >= 3
        # ---- output side -------------------------------------------------------------------------------------
        state.upload(psi0)
        buf0.init_zero(False)
        buf1.init_zero(False)
        passes = state.apply_ops_io(ops, dst=(buf0, bits, buf1 if own >= 0 else None, own))
        g0, g1 = buf0.download(), buf1.download()
        for d, w in enumerate(_slabs(want, bits)):
            got = (g1 if d == own else g0)[d * slab:(d + 1) * slab]
            np.testing.assert_allclose(got, w, rtol=0, atol=1e-11, err_msg=f"k={k} trial={trial} bits={bits} own={own} slab {d}")
        if own >= 0:                                          # nothing else lands in the receive buffer
            other = np.delete(g1, np.s_[own * slab:(own + 1) * slab])
            assert not np.any(other)
        # (with an own slab the fusion needs the slab bits outside the last pass's tile: not known here -- the exact
        # counts are asserted in test_pass_counts_prove_the_fusion)
        if k >= 8:
            assert passes == plain_passes + (0 if fusable and own < 0 else 1) or (fusable and passes == plain_passes), \
                (k, trial, bits, own, passes, plain_passes)
        # ---- input side --------------------------------------------------------------------------------------
        buf1.upload(np.concatenate(_slabs(psi0, bits)))
        state.init_zero(False)
        passes = state.apply_ops_io(ops, src=(buf1, bits))
        np.testing.assert_allclose(state.download(), want, rtol=0, atol=1e-11, err_msg=f"k={k} trial={trial} bits={bits} (input)")
        if k >= 8:
            assert passes == plain_passes + (0 if fusable else 1), (k, trial, bits, passes, plain_passes)
        # ---- both ends at once (a second, different slab layout on the way out) ------------------------------
        bits2 = [int(b) for b in rng.choice(np.arange(3, k), size=min(m, k - 3), replace=False)]
        buf1.upload(np.concatenate(_slabs(psi0, bits)))
        buf0.init_zero(False)
        state.init_zero(False)
        state.apply_ops_io(ops, src=(buf1, bits), dst=(buf0, bits2, None, -1))
        np.testing.assert_allclose(buf0.download(), np.concatenate(_slabs(want, bits2)), rtol=0, atol=1e-11,
                                   err_msg=f"k={k} trial={trial} {bits}->{bits2}")
    for c in (state, buf0, buf1):
        c.close()


def test_pass_counts_prove_the_fusion():
    """Slab bits at the top of the index, ops on the qubits below them: the slab bits are never tile bits, so both ends
    fuse and the op list costs exactly the passes it costs in place -- two fewer than pack + passes + unpack."""
    from quantum_simulations_amd.kernel.device import DeviceChunk
    k = 20
    state, buf0, buf1, buf2 = DeviceChunk.empty(k), DeviceChunk.empty(k), DeviceChunk.empty(k), DeviceChunk.empty(k)
    for m, own in ((1, 0), (2, 3), (3, 5), (2, -1)):
        bits = list(range(k - m, k))[::-1]
        ops = _random_ops(k - 3, 120, 4400 + m)
        psi0 = _rand_state(k, 4500 + m)
        want = psi0.copy()
        orc.apply_ops(want, ops)
        state.upload(psi0)
        plain_passes = state.apply_ops(ops)
        assert plain_passes >= 2
        buf2.upload(np.concatenate(_slabs(psi0, bits)))       # (the source must be a third buffer: one pass may read
        buf0.init_zero(False)                                 # it while it writes the destinations)
        state.init_zero(False)
        passes = state.apply_ops_io(ops, src=(buf2, bits), dst=(buf0, bits, buf1 if own >= 0 else None, own))
        assert passes == plain_passes, (m, own, passes, plain_passes)
        with pytest.raises(ValueError, match="source buffer must differ"):
            state.apply_ops_io(ops, src=(buf0, bits), dst=(buf0, bits, buf1, max(own, 0)))
        slab = 1 << (k - m)
        g0, g1 = buf0.download(), buf1.download()
        for d, w in enumerate(_slabs(want, bits)):
            np.testing.assert_allclose((g1 if d == own else g0)[d * slab:(d + 1) * slab], w, rtol=0, atol=1e-11, err_msg=f"m={m} slab {d}")
    for c in (state, buf0, buf1, buf2):
        c.close()


@pytest.mark.parametrize("split", [False, True])
def test_three_buffers_own_slab_in_the_source_buffer_or_in_the_chunk(split):
    """dst_own == src (a rank then holds three shard-sized buffers, not four): with two or more kernels the own slab lands
    in the consumed source buffer; when ONE pass reads the source and stores the slabs it lands in the chunk itself and
    qsim_apply_ops_io_own_slab says so.  Whole and split forms (pieces stored / announced in random order)."""
    from quantum_simulations_amd.kernel.device import DeviceChunk
    k = 20
    rng = np.random.default_rng(77 + split)
    state, buf0, buf1 = DeviceChunk.empty(k), DeviceChunk.empty(k), DeviceChunk.empty(k)
    seen = set()
    for case, (m_in, m_out, n_ops) in enumerate(((2, 2, 120), (2, 1, 3), (1, 3, 4), (3, 2, 90), (1, 1, 0), (2, 2, 1))):
        bits_in = [int(b) for b in rng.choice(np.arange(3, k), size=m_in, replace=False)]
        bits_out = list(range(k - m_out, k))[::-1]            # (at the top: never tile bits of a pass on the qubits below)
        own = int(rng.integers(0, 1 << m_out))
        ops = _random_ops(k - 3, n_ops, 9100 + case) if n_ops else []
        psi0 = _rand_state(k, 9200 + case)
        want = psi0.copy()
        orc.apply_ops(want, ops)
        buf1.upload(np.concatenate(_slabs(psi0, bits_in)))
        buf0.init_zero(False)
        state.init_zero(False)
        passes = state.apply_ops_io(ops, src=(buf1, bits_in), dst=(buf0, bits_out, buf1, own),
                                    parts=-4 if split else 0, src_parts=-4 if split else 0)
        if split:
            n_src = state.source_parts()[0]
            for j in rng.permutation(n_src):
                state.load_part(int(j))
            for j in rng.permutation(len(state.pending_parts())):
                state.store_part(int(j))
        in_chunk = state.own_slab_in_chunk()
        assert in_chunk == (passes == 1), (case, passes, in_chunk)
        seen.add(in_chunk)
        slab = 1 << (k - m_out)
        g0, g1 = buf0.download(), (state if in_chunk else buf1).download()
        for d, w in enumerate(_slabs(want, bits_out)):
            np.testing.assert_allclose((g1 if d == own else g0)[d * slab:(d + 1) * slab], w, rtol=0, atol=1e-11,
                                       err_msg=f"case {case}: {bits_in} -> {bits_out}, own {own} (in chunk: {in_chunk}), slab {d}")
    assert seen == {False, True}
    for c in (state, buf0, buf1):
        c.close()


def test_single_pass_both_ends_and_argument_checks():
    from quantum_simulations_amd.kernel.device import DeviceChunk
    k = 14
    state, buf0, buf1 = DeviceChunk.empty(k), DeviceChunk.empty(k), DeviceChunk.empty(k)
    psi0 = _rand_state(k, 5)
    ops = [([5], orc.gate_matrix("H")), ([5, 9], orc.gate_matrix("CNOT")), ([2], orc.gate_matrix("T"))]
    want = psi0.copy()
    orc.apply_ops(want, ops)
    buf2 = DeviceChunk.from_numpy(np.concatenate(_slabs(psi0, [12, 4])))
    assert state.apply_ops_io(ops, src=(buf2, [12, 4]), dst=(buf0, [13], buf1, 1)) == 1      # ONE pass does it all
    np.testing.assert_allclose(buf0.download()[:1 << (k - 1)], _slabs(want, [13])[0], rtol=0, atol=1e-12)
    np.testing.assert_allclose(buf1.download()[1 << (k - 1):], _slabs(want, [13])[1], rtol=0, atol=1e-12)
    buf2.close()
    with pytest.raises(ValueError):
        state.apply_ops_io(ops, dst=(state, [5], None, -1))          # destination must be another buffer
    with pytest.raises(ValueError):
        state.apply_ops_io(ops, dst=(buf0, [5, 5], None, -1))
    with pytest.raises(NotImplementedError, match="non-local"):
        state.apply_ops_io(ops, src=(buf1, [k]))
    with pytest.raises(ValueError):
        state.apply_ops_io(ops, dst=(buf0, [5], buf0, 0))            # own buffer = destination
    small = DeviceChunk.empty(10)
    with pytest.raises(ValueError):
        state.apply_ops_io(ops, src=(small, [5]))
    # qsim_ops_io carries its size; a host built against another layout of the struct (a shorter, older one; a
    # zeroed one) is refused with a message that says what to do, instead of being read past its end
    import ctypes as C

    from quantum_simulations_amd import _lib
    from quantum_simulations_amd.kernel.device import pack_ops
    nq, qs, mats = pack_ops(ops)
    for bad_size in (0, C.sizeof(_lib.OpsIo) - 16, C.sizeof(_lib.OpsIo) + 8):
        io = _lib.OpsIo()
        io.struct_size = bad_size
        io.dst, io.dst_m, io.own_pattern = buf0._h, 1, -1
        io.dst_bits[0] = 13
        with pytest.raises(ValueError, match="struct_size"):
            _lib.check(_lib.load().qsim_apply_ops_io(state._h, len(nq), nq.ctypes.data_as(C.c_void_p), qs.ctypes.data_as(C.c_void_p),
                                                     mats.ctypes.data_as(C.c_void_p), C.byref(io), None))
    # ... and the tiles of the first passes may be named through it (any masks: useless ones are ignored)
    state.upload(psi0)
    hints = np.array([(1 << 5) | (1 << 9) | 0b11111 << 6, 1 << 13, 0], dtype=np.uint64)
    assert state.apply_ops_io(ops, dst=(buf0, [13], buf1, 1), tiles=hints) >= 1
    np.testing.assert_allclose(buf0.download()[:1 << (k - 1)], _slabs(want, [13])[0], rtol=0, atol=1e-12)
    for c in (state, buf0, buf1, small):
        c.close()


@pytest.mark.parametrize("k", [14, 17, 21])
def test_split_form_stores_the_slabs_piece_by_piece(k):
    """qsim_ops_io::dst_parts: the slab-storing pass is cut into PIECES (the j-th equal sub-range of every slab; VERDICT r03
    item 6).  The cut depends only on (k, m, pieces asked for) -- whatever the op list, so every rank of a multi-GPU run
    posts the same messages -- while the op list decides how many partial launches the pass takes (piece bits that are
    tile bits of the pass merge launches).  After piece j has been stored, its range of every slab holds the final
    amplitudes; after the last one the buffers equal the one-launch form bit for bit.  Covers fused passes with the top
    bits free / taken by the tile, the pack fallback (slab bit inside a line, empty op list), pieces stored out of order."""
    from quantum_simulations_amd.kernel.device import DeviceChunk, split_piece_count
    from quantum_simulations_amd.runner.distributed import split_pieces
    rng = np.random.default_rng(900 + k)
    state, buf0, buf1, ref0, ref1 = (DeviceChunk.empty(k) for _ in range(5))
    n = 1 << k
    launches_seen = set()
    for trial in range(10):
        m = int(rng.integers(1, 4))
        bits = [int(b) for b in rng.choice(np.arange(3, k), size=m, replace=False)]
        if trial == 7:
            bits[0] = 1                                        # slab bit inside a 128-byte line: pack fallback
        own = int(rng.integers(0, 1 << m)) if trial % 3 else -1
        ops = _random_ops(k, 40, 9100 + 13 * k + trial) if trial != 8 else []
        top = [b for b in range(k - 1, -1, -1) if b not in bits][:2]
        if trial in (2, 5):                                    # gates on the top qubits: the piece bits are tile bits of the last pass
            ops += [([q], orc.gate_matrix("H")) for q in top]
        if trial in (3, 6):                                    # ... or nothing touches them: the pass splits into as many launches as pieces
            ops = [(qs, U) for qs, U in ops if not set(qs) & set(top)]
        psi0 = _rand_state(k, 9200 + trial)
        slab = n >> m
        state.upload(psi0)
        ref0.init_zero(False)
        ref1.init_zero(False)
        state.apply_ops_io(ops, dst=(ref0, bits, ref1 if own >= 0 else None, own))
        want0, want1 = ref0.download(), ref1.download()
        state.upload(psi0)
        buf0.init_zero(False)
        buf1.init_zero(False)
        state.apply_ops_io(ops, dst=(buf0, bits, buf1 if own >= 0 else None, own), parts=-4)
        parts = state.pending_parts()
        assert parts == split_pieces(k, m, -4) and len(parts) == split_piece_count(k, m, -4) == 4     # the same rule everywhere
        launches_seen.add(state.last_split_launches)
        if trial in (3, 6) and min(bits) >= 3:
            assert state.last_split_launches == 4, (trial, state.last_split_launches)
        covered = np.zeros(slab, dtype=bool)
        for j in (int(x) for x in rng.permutation(len(parts))):
            state.store_part(j)
            off, cnt = parts[j]
            assert not covered[off:off + cnt].any()
            covered[off:off + cnt] = True
            g0, g1 = buf0.download(), buf1.download()
            for d in range(1 << m):
                got, want = (g1, want1) if d == own else (g0, want0)
                sl = slice(d * slab, (d + 1) * slab)
                np.testing.assert_array_equal(got[sl][covered], want[sl][covered], err_msg=f"k={k} trial={trial} piece {j} slab {d}")
        assert covered.all()
        np.testing.assert_array_equal(buf0.download(), want0)
        np.testing.assert_array_equal(buf1.download(), want1)
        with pytest.raises(ValueError):
            state.store_part(0)                                # nothing pending any more
    assert {1, 4} <= launches_seen or {2, 4} <= launches_seen, launches_seen
    # the real floor: pieces keep >= 2^20 amplitudes, so these shards are not cut
    state.apply_ops_io(_random_ops(k, 10, 1), dst=(buf0, [5], None, -1), parts=4)
    assert len(state.pending_parts()) == (1 if k - 1 < 21 else 1 << min(2, k - 1 - 20))
    # a second split call while pieces are pending is refused, and so is a plain op list
    with pytest.raises(ValueError, match="pending"):
        state.apply_ops_io(_random_ops(k, 10, 1), dst=(buf0, [5], None, -1), parts=-2)
    with pytest.raises(ValueError, match="pending"):
        state.apply_ops(_random_ops(k, 10, 1))
    for j in range(len(state.pending_parts())):
        state.store_part(j)
    for c in (state, buf0, buf1, ref0, ref1):
        c.close()


@pytest.mark.parametrize("k", [7, 14, 17, 21])
def test_split_source_consumes_the_pieces_as_they_arrive(k):
    """qsim_ops_io::src_parts (the receive side of a fused re-layout): the call plans and launches nothing; every
    `load_part(j)` may only touch source pieces that have been announced.  Here a piece's data is written into the source
    buffer (NaN before) right before it is announced and everything queued is drained before the next one, so a launch
    that read an unannounced piece would leave NaNs.  First pass with the top bits free (partial launches) and taken by
    the tile (whole, with the last piece), the unpack fallback (slab bit inside a line), chunks too small for tiles, an
    empty op list, a destination in one launch and in pieces, a single pass that is also the slab-storing one."""
    from quantum_simulations_amd.kernel.device import DeviceChunk
    from quantum_simulations_amd.runner.distributed import split_pieces
    rng = np.random.default_rng(1300 + k)
    state, src, dst, keep = (DeviceChunk.empty(k) for _ in range(4))
    n = 1 << k
    idx = np.arange(n)
    launches_seen = set()
    for trial in range(12):
        m = int(rng.integers(1, min(3, k - 4) + 1))
        bits = [int(b) for b in rng.choice(np.arange(3, k), size=m, replace=False)]
        if trial == 7:
            bits[0] = 2                                       # slab bit inside a 128-byte line: unpack pieces
        ops = _random_ops(k, 40, 7700 + 13 * k + trial) if trial != 8 else []
        top = [b for b in range(k - 1, -1, -1) if b not in bits][:2]
        if trial in (2, 5):                                   # the piece bits are tile bits of the first pass
            ops = [([q], orc.gate_matrix("H")) for q in top] + ops
        if trial in (3, 6, 9):                                # nothing touches them at all
            ops = [(qs, U) for qs, U in ops if not set(qs) & set(top)]
        if trial == 10:                                       # ONE pass that is also the slab-storing pass
            ops = [([5], orc.gate_matrix("H")), ([5, 6], orc.gate_matrix("CNOT"))]
        with_dst = trial in (4, 6, 10, 11)
        psi0 = _rand_state(k, 7800 + trial)
        want = psi0.copy()
        orc.apply_ops(want, ops)
        slab = n >> m
        pat = sum(((idx >> b) & 1) << i for i, b in enumerate(bits))
        packed = np.concatenate([psi0[pat == d] for d in range(1 << m)])
        src.upload(np.full(n, np.nan + 1j * np.nan))
        state.upload(np.full(n, np.nan + 1j * np.nan))
        bits2 = [int(b) for b in rng.choice(np.arange(3, k), size=m, replace=False)]
        own = int(rng.integers(0, 1 << m))
        dst.init_zero(False)
        keep.init_zero(False)
        passes = state.apply_ops_io(ops, src=(src, bits), dst=(dst, bits2, keep, own) if with_dst else None,
                                    parts=-2 if trial in (6, 10) else 0, src_parts=-4)
        assert passes >= 1
        pieces = split_pieces(k, m, -4)
        n_parts, amps_, launches = state.source_parts()
        assert n_parts == len(pieces) and amps_ == pieces[0][1]
        launches_seen.add(launches)
        with pytest.raises(ValueError, match="pending"):
            state.apply_ops(ops or [([3], orc.gate_matrix("H"))])
        for j in (int(x) for x in rng.permutation(n_parts)):
            off, cnt = pieces[j]
            for d in range(1 << m):
                src.upload(packed[d * slab + off:d * slab + off + cnt], d * slab + off)
            state.load_part(j)
            state.sync()
        if with_dst:
            if trial in (6, 10):
                for j in range(len(state.pending_parts())):
                    state.store_part(j)
            g0, g1 = dst.download(), keep.download()
            pat2 = sum(((idx >> b) & 1) << i for i, b in enumerate(bits2))
            for d in range(1 << m):
                got = (g1 if d == own else g0)[d * slab:(d + 1) * slab]
                np.testing.assert_allclose(got, want[pat2 == d], rtol=0, atol=1e-11, err_msg=f"k={k} trial={trial} slab {d}")
        else:
            np.testing.assert_allclose(state.download(), want, rtol=0, atol=1e-11, err_msg=f"k={k} trial={trial} bits={bits}")
        with pytest.raises(ValueError):
            state.load_part(0)                                # nothing pending any more
    if k >= 14:
        assert 4 in launches_seen and 0 in launches_seen, launches_seen
    for c in (state, src, dst, keep):
        c.close()
