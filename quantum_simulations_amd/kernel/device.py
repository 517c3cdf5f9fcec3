"""HBM-resident chunk handle: the GPU counterpart of the reference's in-RAM numpy chunk
(wenbo_engine/storage/block_store.py:31, chunk c = amplitudes [c*2^k, (c+1)*2^k)).

A `DeviceChunk` owns (or views, or wraps) 2^k complex128 amplitudes on one MI355X and
forwards every operation to libqsim_hip.so.  No arithmetic happens in Python.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from quantum_simulations_amd import _lib


def _mat_ptr(U: np.ndarray, dim: int):
    m = np.ascontiguousarray(U, dtype=np.complex128)
    if m.shape != (dim, dim):
        raise ValueError(f"expected a {dim}x{dim} matrix, got shape {m.shape}")
    return m, m.ctypes.data_as(C.c_void_p)


def pack_ops(ops) -> tuple[np.ndarray, np.ndarray, np.ndarray]:
    """[(qubits, U), ...] -> (nq int32[n], qubits int32[2n], mats float64[32n])."""
    n = len(ops)
    nq = np.zeros(n, dtype=np.int32)
    qs = np.zeros(2 * n, dtype=np.int32)
    mats = np.zeros((n, 16), dtype=np.complex128)
    for i, (qubits, U) in enumerate(ops):
        nq[i] = len(qubits)
        qs[2 * i: 2 * i + len(qubits)] = qubits
        flat = np.asarray(U, dtype=np.complex128).reshape(-1)
        if flat.size != (4 if len(qubits) == 1 else 16):
            raise ValueError(f"op {i}: matrix size {flat.size} does not fit {len(qubits)} qubit(s)")
        mats[i, : flat.size] = flat
    return nq, qs, mats


class DeviceChunk:
    """2^k complex128 amplitudes in HBM."""

    def __init__(self, handle: int, k: int, device: int, keep=None):
        self._h = C.c_void_p(handle)
        self.k = k
        self.device = device
        self._keep = keep  # parent chunk / torch tensor that owns the memory

    # ---- construction -------------------------------------------------------------
    @classmethod
    def empty(cls, k: int, device: int = 0) -> "DeviceChunk":
        lib = _lib.load()
        h = C.c_void_p()
        _lib.check(lib.qsim_create(device, k, C.byref(h)))
        return cls(h.value, k, device)

    @classmethod
    def zero_state(cls, k: int, device: int = 0, set_amp0: bool = True) -> "DeviceChunk":
        c = cls.empty(k, device)
        c.init_zero(set_amp0)
        return c

    @classmethod
    def from_numpy(cls, arr: np.ndarray, device: int = 0) -> "DeviceChunk":
        n = arr.shape[0]
        if arr.ndim != 1 or n < 1 or n & (n - 1):
            raise ValueError("chunk must be a 1-D array with a power-of-two length")
        c = cls.empty(n.bit_length() - 1, device)
        c.upload(arr)
        return c

    @classmethod
    def wrap_pointer(cls, ptr: int, k: int, device: int, stream: int = 0, keep=None) -> "DeviceChunk":
        lib = _lib.load()
        h = C.c_void_p()
        _lib.check(lib.qsim_wrap(device, C.c_void_p(ptr), k, C.c_void_p(stream), C.byref(h)))
        return cls(h.value, k, device, keep=keep)

    def view(self, offset_amps: int, k: int) -> "DeviceChunk":
        lib = _lib.load()
        h = C.c_void_p()
        _lib.check(lib.qsim_create_view(self._h, offset_amps, k, C.byref(h)))
        return DeviceChunk(h.value, k, self.device, keep=self)

    def close(self) -> None:
        if self._h is not None and self._h.value:
            _lib.load().qsim_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):  # best effort
        try:
            self.close()
        except Exception:
            pass

    def __len__(self) -> int:
        return 1 << self.k

    @property
    def device_ptr(self) -> int:
        return _lib.load().qsim_device_ptr(self._h) or 0

    # ---- state I/O ----------------------------------------------------------------
    def init_zero(self, set_amp0: bool = True) -> None:
        _lib.check(_lib.load().qsim_init_zero(self._h, 1 if set_amp0 else 0))

    def init_random(self, seed: int) -> None:
        _lib.check(_lib.load().qsim_init_random(self._h, seed))

    def upload(self, arr: np.ndarray, offset: int = 0) -> None:
        host = np.ascontiguousarray(arr, dtype=np.complex128)
        _lib.check(_lib.load().qsim_upload(self._h, host.ctypes.data_as(C.c_void_p), offset, host.size))

    def download(self, offset: int = 0, count: int | None = None) -> np.ndarray:
        count = len(self) - offset if count is None else count
        out = np.empty(count, dtype=np.complex128)
        _lib.check(_lib.load().qsim_download(self._h, out.ctypes.data_as(C.c_void_p), offset, count))
        return out

    def download_c64(self, offset: int = 0, count: int | None = None) -> np.ndarray:
        """Amplitudes as complex64, rounded on the device (the reference's chunk-file dtype)."""
        count = len(self) - offset if count is None else count
        out = np.empty(count, dtype=np.complex64)
        _lib.check(_lib.load().qsim_download_c64(self._h, out.ctypes.data_as(C.c_void_p), int(offset), int(count)))
        return out

    def upload_c64(self, arr: np.ndarray, offset: int = 0) -> None:
        a = np.ascontiguousarray(arr, dtype=np.complex64)
        _lib.check(_lib.load().qsim_upload_c64(self._h, a.ctypes.data_as(C.c_void_p), int(offset), int(a.size)))

    def copy_from(self, other: "DeviceChunk", variant: int = 0) -> None:
        """variant (measurement aid, qsim_copy_variant): 0 the library's choice, 1 non-temporal kernel, 2 plain kernel, 3 hipMemcpyAsync"""
        if variant:
            _lib.check(_lib.load().qsim_copy_variant(self._h, other._h, int(variant)))
        else:
            _lib.check(_lib.load().qsim_copy(self._h, other._h))

    # ---- gates --------------------------------------------------------------------
    def apply_1q(self, qubit: int, U: np.ndarray) -> None:
        m, p = _mat_ptr(U, 2)
        _lib.check(_lib.load().qsim_apply_1q(self._h, int(qubit), p))

    def apply_2q(self, qa: int, qb: int, U: np.ndarray) -> None:
        m, p = _mat_ptr(U, 4)
        _lib.check(_lib.load().qsim_apply_2q(self._h, int(qa), int(qb), p))

    def apply_fused_k(self, qubits, M: np.ndarray) -> None:
        """Dense k-qubit block (qsim_apply_fused_k, k <= 6): M is 2^k x 2^k, M[out, in], pattern bit i <-> qubits[i] --
        v3's `_apply_combined_matrix` (parallel_gate_applicator.py:315-385) on the dense state."""
        q = np.asarray(qubits, dtype=np.int32)
        m = np.ascontiguousarray(M, dtype=np.complex128)
        if m.shape != (1 << len(q), 1 << len(q)):
            raise ValueError(f"a {len(q)}-qubit block needs a {1 << len(q)} x {1 << len(q)} matrix, got {m.shape}")
        _lib.check(_lib.load().qsim_apply_fused_k(self._h, len(q), q.ctypes.data_as(C.c_void_p), m.ctypes.data_as(C.c_void_p)))

    def apply_ops_tiled(self, ops, tile_masks) -> int:
        """`apply_ops` (fused) with the high tile bits of the first passes given (qsim_apply_ops_tiled): `tile_masks` =
        uint64 array, bit b of entry p = index bit b is a tile bit of pass p.  Returns the HBM round trips."""
        nq, qs, mats = ops if (isinstance(ops, tuple) and len(ops) == 3 and isinstance(ops[0], np.ndarray)) else pack_ops(ops)
        tm = np.ascontiguousarray(tile_masks, dtype=np.uint64)
        lib = _lib.load()
        _lib.check(lib.qsim_apply_ops_tiled(self._h, len(nq), nq.ctypes.data_as(C.c_void_p), qs.ctypes.data_as(C.c_void_p),
                                            mats.ctypes.data_as(C.c_void_p), len(tm), tm.ctypes.data_as(C.c_void_p)))
        return lib.qsim_last_pass_count(self._h)

    def apply_ops(self, ops, fused: bool = True) -> int:
        """One pass: every (qubits, U) of `ops` in one C call.  `fused` groups them into LDS-tile
        launches (order kept for ops sharing a qubit); returns the number of HBM round trips.
        `ops` may also be the tuple returned by `pack_ops` (plans that run repeatedly pack once)."""
        if not len(ops):
            return 0
        if isinstance(ops, tuple) and len(ops) == 3 and isinstance(ops[0], np.ndarray):
            nq, qs, mats = ops
            ops = nq
        else:
            nq, qs, mats = pack_ops(ops)
        lib = _lib.load()
        fn = lib.qsim_apply_ops if fused else lib.qsim_apply_ops_unfused
        _lib.check(fn(self._h, len(ops), nq.ctypes.data_as(C.c_void_p), qs.ctypes.data_as(C.c_void_p),
                      mats.ctypes.data_as(C.c_void_p)))
        return lib.qsim_last_pass_count(self._h) if fused else len(ops)

    # ---- sync / reductions / timing -------------------------------------------------
    def apply_ops_io(self, ops, src=None, dst=None, parts: int = 0, src_parts: int = 0, tiles=None) -> int:
        """`apply_ops` with a re-layout fused into its ends (qsim_apply_ops_io): `src` = (chunk, bits): the state
        is read from that chunk in the slab layout of `pack_all` over `bits`; `dst` = (chunk, bits, own_chunk,
        own_pattern): it is left in `chunk` in slab layout, slab `own_pattern` (>= 0) in `own_chunk` -- which may be the
        source chunk: see `own_slab_in_chunk`.  Returns the HBM passes made.  `parts` = 2, 4, 8 (with `dst`): split form -- the slabs are NOT stored by this call but piece
        by piece by `store_part(j)` for every j < len(`pending_parts()`), so that each piece's exchange can be posted
        while the later pieces are still computed (negative: no 2^20-amplitude floor on a piece, for tests).
        `src_parts` (with `src`): the source arrives in the pieces of the same rule: this call only plans; announce every
        piece with `load_part(j)` once its transfer is ordered on this chunk's stream -- the first pass starts on the
        tiles whose pieces are there, the rest runs with the last piece.  `tiles`: uint64 masks, the high tile bits of the
        first passes named by the caller (qsim_ops_io::tile_masks; as in `apply_ops_tiled`)."""
        nq, qs, mats = pack_ops(ops) if not (isinstance(ops, tuple) and len(ops) == 3 and isinstance(ops[0], np.ndarray)) else ops
        io = _lib.OpsIo()
        keep = []
        if tiles is not None and len(tiles):
            tm = np.ascontiguousarray(tiles, dtype=np.uint64)
            io.n_tiles, io.tile_masks = len(tm), tm.ctypes.data
            keep.append(tm)
        if src is not None:
            chunk, bits = src
            io.src, io.src_m = chunk._h, len(bits)
            for i, b in enumerate(bits):
                io.src_bits[i] = int(b)
            keep.append(chunk)
        io.own_pattern = -1
        if dst is not None:
            chunk, bits, own_chunk, own_pattern = dst
            io.dst, io.dst_m = chunk._h, len(bits)
            for i, b in enumerate(bits):
                io.dst_bits[i] = int(b)
            if own_chunk is not None and own_pattern >= 0:
                io.dst_own, io.own_pattern = own_chunk._h, int(own_pattern)
            keep += [chunk, own_chunk]
        io.dst_parts = int(parts) if dst is not None else 0
        io.src_parts = int(src_parts) if src is not None else 0
        passes = C.c_int()
        _lib.check(_lib.load().qsim_apply_ops_io(self._h, len(nq), nq.ctypes.data_as(C.c_void_p), qs.ctypes.data_as(C.c_void_p),
                                                 mats.ctypes.data_as(C.c_void_p), C.byref(io), C.byref(passes)))
        return passes.value

    def own_slab_in_chunk(self) -> bool:
        """True when the last `apply_ops_io` left the slab that stays on the rank in THIS chunk instead of `own_chunk`
        (own_chunk was the source chunk and one pass did everything; qsim_apply_ops_io_own_slab)."""
        flag = C.c_int32()
        _lib.check(_lib.load().qsim_apply_ops_io_own_slab(self._h, C.byref(flag)))
        return bool(flag.value)

    def pending_parts(self) -> list:
        """Pieces of the pending split `apply_ops_io`: [(offset, amplitudes), ...] -- piece j of EVERY slab d is that
        range of [d * slab, (d + 1) * slab) in the send / receive buffers (qsim_apply_ops_io_parts).  The cut depends only
        on (k, m, parts asked for): every rank gets the same list."""
        n_parts, amps_, launches = C.c_int32(), C.c_uint64(), C.c_int32()
        _lib.check(_lib.load().qsim_apply_ops_io_parts(self._h, C.byref(n_parts), C.byref(amps_), C.byref(launches)))
        self.last_split_launches = launches.value
        return [(j * int(amps_.value), int(amps_.value)) for j in range(n_parts.value)]

    def store_part(self, part: int) -> None:
        _lib.check(_lib.load().qsim_apply_ops_io_part(self._h, int(part)))

    def source_parts(self) -> tuple:
        """(pieces, amplitudes per piece, partial launches of the first pass) of the pending op list with a split source."""
        n_parts, amps_, launches = C.c_int32(), C.c_uint64(), C.c_int32()
        _lib.check(_lib.load().qsim_apply_ops_io_source_parts(self._h, C.byref(n_parts), C.byref(amps_), C.byref(launches)))
        return n_parts.value, int(amps_.value), launches.value

    def load_part(self, part: int) -> None:
        """Piece `part` of the source of the pending op list has arrived (qsim_apply_ops_io_load)."""
        _lib.check(_lib.load().qsim_apply_ops_io_load(self._h, int(part)))

    def sync(self) -> None:
        _lib.check(_lib.load().qsim_sync(self._h))

    def count_nonzero(self, eps: float = 1e-15) -> int:
        """Amplitudes with |re| > eps or |im| > eps (v3's pruning rule), counted on the device."""
        n = C.c_uint64()
        _lib.check(_lib.load().qsim_count_nonzero(self._h, float(eps), C.byref(n)))
        return int(n.value)

    def export_nonzero(self, eps: float = 1e-15, capacity: int | None = None):
        """(indices uint64[m], amplitudes complex128[m]) of the kept amplitudes, ascending by index, selected on the
        device (no dense download); None when there are more than `capacity` of them."""
        if capacity is None:
            capacity = self.count_nonzero(eps)
        idx = np.empty(capacity, dtype=np.uint64)
        amp = np.empty(capacity, dtype=np.complex128)
        n = C.c_uint64()
        _lib.check(_lib.load().qsim_export_nonzero(self._h, float(eps), int(capacity), idx.ctypes.data_as(C.c_void_p),
                                                   amp.ctypes.data_as(C.c_void_p), C.byref(n)))
        if n.value > capacity:
            return None
        return idx[:n.value], amp[:n.value]

    def norm2(self) -> float:
        out = C.c_double()
        _lib.check(_lib.load().qsim_norm2(self._h, C.byref(out)))
        return out.value

    def max_abs_err_closed_form(self, kind: str, n_total: int, base_index: int = 0,
                                log_to_phys=None) -> float:
        """max |amp - closed form| over this chunk; `log_to_phys` maps a staged layout back."""
        out = C.c_double()
        code = {"ghz": 0, "ghz_qft": 1}[kind]
        perm = None if log_to_phys is None else np.asarray(log_to_phys, dtype=np.int32)
        _lib.check(_lib.load().qsim_max_abs_err_closed_form_perm(
            self._h, code, n_total, base_index,
            None if perm is None else perm.ctypes.data_as(C.c_void_p), C.byref(out)))
        return out.value

    def fingerprint(self, n_total: int, base_index: int = 0, log_to_phys=None, seed: int = 0,
                    sel_mask: int = 0, sel_value: int = 0) -> complex:
        """sum of amp * w(logical index) over this chunk's amplitudes whose logical index y has (y & sel_mask) ==
        sel_value (qsim_fingerprint: counter-based weights, layout-aware, evaluated on the device)."""
        out = (C.c_double * 2)()
        perm = None if log_to_phys is None else np.asarray(log_to_phys, dtype=np.int32)
        _lib.check(_lib.load().qsim_fingerprint(
            self._h, n_total, base_index, None if perm is None else perm.ctypes.data_as(C.c_void_p),
            seed, sel_mask, sel_value, out))
        return complex(out[0], out[1])

    def time_begin(self) -> None:
        _lib.check(_lib.load().qsim_time_begin(self._h))

    def time_end(self) -> float:
        ms = C.c_float()
        _lib.check(_lib.load().qsim_time_end(self._h, C.byref(ms)))
        return ms.value

    def profile_begin(self) -> None:
        """Bracket every gate launch on this chunk's stream with HIP events until profile_end."""
        _lib.check(_lib.load().qsim_profile_begin(self._h))

    def profile_end(self) -> list[dict]:
        entries = (_lib.ProfileEntry * 16)()
        n = C.c_int()
        _lib.check(_lib.load().qsim_profile_end(self._h, 16, C.byref(n), C.cast(entries, C.c_void_p)))
        out = []
        for e in entries[: min(n.value, 16)]:
            out.append({"kernel": e.kernel.decode(), "launches": int(e.launches),
                        "total_ms": float(e.total_ms), "algorithmic_bytes": float(e.algorithmic_bytes),
                        "hbm_bytes": float(e.hbm_bytes), "streaming_launches": int(e.streaming_launches)})
        return out

    def pack_half(self, bit: int, value: int, buf: "DeviceChunk") -> None:
        _lib.check(_lib.load().qsim_pack_half(self._h, bit, value, buf._h))

    def unpack_half(self, bit: int, value: int, buf: "DeviceChunk") -> None:
        _lib.check(_lib.load().qsim_unpack_half(self._h, bit, value, buf._h))


    def pack_bits(self, bits, pattern: int, buf: "DeviceChunk", buf_offset: int = 0) -> None:
        b = np.asarray(bits, dtype=np.int32)
        _lib.check(_lib.load().qsim_pack_bits(self._h, len(b), b.ctypes.data_as(C.c_void_p), int(pattern),
                                              buf._h, int(buf_offset)))

    def unpack_bits(self, bits, pattern: int, buf: "DeviceChunk", buf_offset: int = 0) -> None:
        b = np.asarray(bits, dtype=np.int32)
        _lib.check(_lib.load().qsim_unpack_bits(self._h, len(b), b.ctypes.data_as(C.c_void_p), int(pattern),
                                                buf._h, int(buf_offset)))
    def pack_all(self, bits, buf: "DeviceChunk", skip_pattern: int = -1, piece: int = 0, n_pieces: int = 1) -> None:
        """Every slab of the all-to-all re-layout in one pass (qsim_pack_all); `piece` of `n_pieces`
        = the same contiguous sub-range of every slab."""
        b = np.asarray(bits, dtype=np.int32)
        _lib.check(_lib.load().qsim_pack_all(self._h, len(b), b.ctypes.data_as(C.c_void_p), buf._h, int(skip_pattern),
                                             int(piece), int(n_pieces)))

    def unpack_all(self, bits, buf: "DeviceChunk", skip_pattern: int = -1, piece: int = 0, n_pieces: int = 1) -> None:
        b = np.asarray(bits, dtype=np.int32)
        _lib.check(_lib.load().qsim_unpack_all(self._h, len(b), b.ctypes.data_as(C.c_void_p), buf._h, int(skip_pattern),
                                               int(piece), int(n_pieces)))


def split_piece_count(k: int, m: int, parts: int) -> int:
    """Pieces the split form of `apply_ops_io` cuts every slab into (qsim_split_piece_count: a pure function of the chunk
    size, the slab bits' count and the pieces asked for -- no device needed)."""
    return int(_lib.load().qsim_split_piece_count(int(k), int(m), int(parts)))


def device_count() -> int:
    n = C.c_int()
    _lib.check(_lib.load().qsim_device_count(C.byref(n)))
    return n.value


def relayout_plan(rank: int, world: int, k: int, local_bits, global_bits, n_pieces: int = 4) -> dict:
    """The schedule of `Comm.relayout` as a pure function of (rank, world, ...) -- qsim_comm_relayout_plan: no GPU and
    no communicator needed.  Returns pieces, own_pattern, peers, slab_offsets (amplitudes, send = receive), piece_amps."""
    lb, gb = np.asarray(local_bits, dtype=np.int32), np.asarray(global_bits, dtype=np.int32)
    n_p, n_peers, own = C.c_int32(), C.c_int32(), C.c_int32()
    peers, offs, part = np.zeros(7, dtype=np.int32), np.zeros(7, dtype=np.uint64), C.c_uint64()
    _lib.check(_lib.load().qsim_comm_relayout_plan(int(rank), int(world), int(k), len(lb), lb.ctypes.data_as(C.c_void_p),
                                                   gb.ctypes.data_as(C.c_void_p), int(n_pieces), C.byref(n_p), C.byref(n_peers),
                                                   C.byref(own), peers.ctypes.data_as(C.c_void_p),
                                                   offs.ctypes.data_as(C.c_void_p), C.byref(part)))
    return {"pieces": n_p.value, "own_pattern": own.value, "peers": [int(x) for x in peers[:n_peers.value]],
            "slab_offsets": [int(x) for x in offs[:n_peers.value]], "piece_amps": int(part.value)}


class Comm:
    """RCCL communicator inside libqsim_hip.so (include/qsim_hip.h, multi-GPU reach of the C ABI): one per
    process / GPU.  `unique_id()` on rank 0, hand the 128 bytes to every rank, then `Comm(device, rank,
    world, uid)`.  The Python runner (runner/distributed.py) uses torch.distributed instead; this wrapper
    is what a host without torch binds."""

    ID_BYTES = 128

    def __init__(self, device: int, rank: int, world: int, uid: bytes):
        if len(uid) != self.ID_BYTES:
            raise ValueError("the unique id has 128 bytes")
        self._h = C.c_void_p()
        buf = (C.c_uint8 * self.ID_BYTES).from_buffer_copy(uid)
        _lib.check(_lib.load().qsim_comm_init(device, rank, world, C.cast(buf, C.c_void_p), C.byref(self._h)))
        self.rank, self.world = rank, world

    @staticmethod
    def unique_id() -> bytes:
        buf = (C.c_uint8 * Comm.ID_BYTES)()
        _lib.check(_lib.load().qsim_comm_get_unique_id(C.cast(buf, C.c_void_p)))
        return bytes(buf)

    def exchange(self, peers, send: DeviceChunk, send_off, recv: DeviceChunk, recv_off, count: int) -> None:
        p = np.asarray(peers, dtype=np.int32)
        so, ro = np.asarray(send_off, dtype=np.uint64), np.asarray(recv_off, dtype=np.uint64)
        _lib.check(_lib.load().qsim_comm_exchange(self._h, len(p), p.ctypes.data_as(C.c_void_p), send._h,
                                                  so.ctypes.data_as(C.c_void_p), recv._h,
                                                  ro.ctypes.data_as(C.c_void_p), int(count)))

    def exchange_bg(self, peers, send: DeviceChunk, send_off, recv: DeviceChunk, recv_off, count: int) -> int:
        """qsim_comm_exchange_bg: the group on the communicator's transfer stream, beside later work on the chunks' stream."""
        p = np.asarray(peers, dtype=np.int32)
        so, ro = np.asarray(send_off, dtype=np.uint64), np.asarray(recv_off, dtype=np.uint64)
        ticket = C.c_uint32()
        _lib.check(_lib.load().qsim_comm_exchange_bg(self._h, len(p), p.ctypes.data_as(C.c_void_p), send._h,
                                                     so.ctypes.data_as(C.c_void_p), recv._h,
                                                     ro.ctypes.data_as(C.c_void_p), int(count), C.byref(ticket)))
        return ticket.value

    def join(self, chunk: DeviceChunk) -> None:
        _lib.check(_lib.load().qsim_comm_join(self._h, chunk._h))

    def wait(self, chunk: DeviceChunk, ticket: int) -> None:
        """The chunk's stream waits for background exchange `ticket` (qsim_comm_wait), not for later ones."""
        _lib.check(_lib.load().qsim_comm_wait(self._h, chunk._h, int(ticket)))

    def relayout(self, state: DeviceChunk, buf0: DeviceChunk, buf1: DeviceChunk, local_bits, global_bits,
                 n_pieces: int = 4) -> None:
        lb, gb = np.asarray(local_bits, dtype=np.int32), np.asarray(global_bits, dtype=np.int32)
        _lib.check(_lib.load().qsim_comm_relayout(self._h, state._h, buf0._h, buf1._h, len(lb),
                                                  lb.ctypes.data_as(C.c_void_p), gb.ctypes.data_as(C.c_void_p),
                                                  int(n_pieces)))

    def relayout_loopback(self, state: DeviceChunk, buf0: DeviceChunk, buf1: DeviceChunk, local_bits, global_bits,
                          n_pieces: int, as_rank: int, as_world: int) -> None:
        """qsim_comm_relayout's pipeline as rank `as_rank` of `as_world` would run it, every transfer looped back
        to this rank (one GPU): the state is unchanged afterwards, buf1 holds the 'received' slabs."""
        lb, gb = np.asarray(local_bits, dtype=np.int32), np.asarray(global_bits, dtype=np.int32)
        _lib.check(_lib.load().qsim_comm_relayout_loopback(self._h, state._h, buf0._h, buf1._h, len(lb),
                                                           lb.ctypes.data_as(C.c_void_p), gb.ctypes.data_as(C.c_void_p),
                                                           int(n_pieces), int(as_rank), int(as_world)))

    def relayout_fused(self, shard: DeviceChunk, send: DeviceChunk, recv: DeviceChunk, before, after, local_bits,
                       global_bits, n_pieces: int = 4, as_rank: int = 0, as_world: int = 0) -> int:
        """qsim_comm_relayout_fused: shard := after(re-layout(before(shard))) with the slabs stored piece by piece by the
        last pass of `before`, exchanged while the next piece is computed, and loaded by the first pass of `after`.
        `before` / `after`: [(qubits, U)] (may be empty).  as_world != 0: loopback form on one GPU.  Returns HBM passes."""
        lb, gb = np.asarray(local_bits, dtype=np.int32), np.asarray(global_bits, dtype=np.int32)
        keep, lists = [], []
        for ops in (before, after):
            ol = _lib.OpList()
            if ops:
                nq, qs, mats = pack_ops(ops)
                keep.append((nq, qs, mats))
                ol.n_ops, ol.nq, ol.qubits, ol.mats = len(nq), nq.ctypes.data, qs.ctypes.data, mats.ctypes.data
            lists.append(ol)
        passes = C.c_int()
        _lib.check(_lib.load().qsim_comm_relayout_fused(self._h, shard._h, send._h, recv._h, C.byref(lists[0]), C.byref(lists[1]),
                                                        len(lb), lb.ctypes.data_as(C.c_void_p), gb.ctypes.data_as(C.c_void_p),
                                                        int(n_pieces), int(as_rank), int(as_world), C.byref(passes)))
        return passes.value

    def apply_2q_quad_remote(self, shard: DeviceChunk, buf: DeviceChunk, ranks, my_index: int, U) -> None:
        """cpu_nonlocal.apply_2q_quad with the four chunks on the ranks `ranks` (chunk j = 2 bit(qa) + bit(qb)); this
        rank holds chunk `my_index`.  ranks = [r] * 4 with r = this rank: one-GPU loopback form."""
        m, p = _mat_ptr(U, 4)
        r = np.asarray(ranks, dtype=np.int32)
        _lib.check(_lib.load().qsim_apply_2q_quad_remote(self._h, shard._h, buf._h, r.ctypes.data_as(C.c_void_p), int(my_index), p))

    def apply_1q_pair_remote(self, shard: DeviceChunk, buf: DeviceChunk, partner: int, my_side: int, U) -> None:
        m, p = _mat_ptr(U, 2)
        _lib.check(_lib.load().qsim_apply_1q_pair_remote(self._h, shard._h, buf._h, int(partner), int(my_side), p))

    def apply_2q_pair_qa_local_remote(self, shard, buf, partner: int, my_side: int, qa: int, U) -> None:
        m, p = _mat_ptr(U, 4)
        _lib.check(_lib.load().qsim_apply_2q_pair_qa_local_remote(self._h, shard._h, buf._h, int(partner),
                                                                  int(my_side), int(qa), p))

    def apply_2q_pair_qb_local_remote(self, shard, buf, partner: int, my_side: int, qb: int, U) -> None:
        m, p = _mat_ptr(U, 4)
        _lib.check(_lib.load().qsim_apply_2q_pair_qb_local_remote(self._h, shard._h, buf._h, int(partner),
                                                                  int(my_side), int(qb), p))

    def close(self) -> None:
        if self._h is not None and self._h.value:
            _lib.load().qsim_comm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
