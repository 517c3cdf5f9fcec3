"""Multi-GPU engine: the 2^n amplitude vector partitioned by its high qubit bits, one process
per GPU (`torch.distributed`; backend "nccl" = RCCL over xGMI on the GPU node, "gloo" in the
CPU tests).

Rank g holds amplitudes [g*2^k, (g+1)*2^k): the reference's chunk g (block_store.py:14-15),
so a gate on qubit q >= k pairs rank g with g XOR 2^(q-k) exactly like partner chunks
(cpu_nonlocal.py:7-15, single_node.py:271-321).  This module replaces the Spark/HiSVSIM chunk
partitioner and the driver-side sequential partner-group loop (spark_runner.py:148-194):

  local gates ....................... HIP kernels on the shard, no communication
  diagonal gate, global qubit(s) .... phase chosen by the rank's own bits, NO exchange
                                       (the reference runs these as butterflies, staging.py:67-72)
  controlled gate, global control ... ranks whose control bit is 1 apply the 1q gate (locally,
                                       or with one partner when the target is global too)
  (local ops produced by the two rows above are queued and fused into the next local pass)
  other gates on a global qubit ..... swap-and-stay: the global qubit trades places with a local one
                                       that is not needed soon (HALF a shard over one xGMI link for one
                                       global qubit, 3/4 over three links for two), the gate then runs
                                       locally in the next fused pass and the layout change is tracked
                                       (the arithmetic of cpu_nonlocal.py:22-67 / single_node.py:271-321
                                       without shipping whole shards there and results back)
  staging SWAP lists ([p_out<k, p_in>=k], SWAP; staging.py:136-152) of one step are MERGED
  into ONE all-to-all re-layout: each rank packs 2^m - 1 slabs, exchanges them with 2^m - 1
  peers concurrently (all links busy) and unpacks in place.

The communication schedule lives here, in Python, and is backend-agnostic; all arithmetic on
amplitudes is done by a `ShardBackend` (HIP: `HipShardBackend`; the CPU test double lives in
tests/ and is never selected by the product).
"""
from __future__ import annotations

import numpy as np

from quantum_simulations_amd.circuit.fusion import batch_levels
from quantum_simulations_amd.circuit.io import levelize, validate_circuit_dict
from quantum_simulations_amd.circuit.staging import atlas_stages, permute_state
from quantum_simulations_amd.kernel import gates as gate_table

_SWAP = gate_table.SWAP()
_I2 = np.eye(2, dtype=np.complex128)


# ------------------------------------------------------------------ gate structure tests
def _is_diagonal(U: np.ndarray) -> bool:
    return not np.any(U - np.diag(np.diag(U)))


def _controlled_on_first(U: np.ndarray):
    """4x4 = |0><0| x I + |1><1| x V (control = qubits[0]) -> V, else None."""
    if np.array_equal(U[:2, :2], _I2) and not np.any(U[:2, 2:]) and not np.any(U[2:, :2]):
        return U[2:, 2:].copy()
    return None


def _controlled_on_second(U: np.ndarray):
    """Control = qubits[1] (identity on pair indices {0,2}, V on {1,3}) -> V, else None."""
    P = U[np.ix_([0, 2, 1, 3], [0, 2, 1, 3])]
    return _controlled_on_first(P)


class HipShardBackend:
    """Shard + exchange buffers as torch CUDA tensors, arithmetic through libqsim_hip.so."""

    def __init__(self, k: int, device: int):
        import torch

        from quantum_simulations_amd.kernel.device import DeviceChunk
        self.torch, self.k, self.device = torch, k, device
        torch.cuda.set_device(device)
        self._DeviceChunk = DeviceChunk
        self._tensors: dict[str, object] = {}
        self._chunks: dict[str, object] = {}
        self.tensor("state")

    def tensor(self, name: str):
        if name not in self._tensors:
            t = self.torch.empty(2 << self.k, dtype=self.torch.float64, device=f"cuda:{self.device}")
            stream = self.torch.cuda.current_stream(self.device).cuda_stream
            self._tensors[name] = t
            self._chunks[name] = self._DeviceChunk.wrap_pointer(t.data_ptr(), self.k, self.device,
                                                                stream=stream, keep=t)
        return self._tensors[name]

    def chunk(self, name: str):
        self.tensor(name)
        return self._chunks[name]

    # ---- state ---------------------------------------------------------------------
    def init_zero(self, set_amp0: bool) -> None:
        self.chunk("state").init_zero(set_amp0)

    def norm2(self) -> float:
        return self.chunk("state").norm2()

    def download(self, offset: int = 0, count: int | None = None) -> np.ndarray:
        return self.chunk("state").download(offset, count)

    def sync(self) -> None:
        self.torch.cuda.synchronize(self.device)

    # ---- arithmetic -----------------------------------------------------------------
    def apply_ops(self, ops, src=None, dst=None, parts: int = 0, src_parts: int = 0, tiles=None) -> int:
        """HBM passes made.  src = (buffer, bits): the shard is read from that buffer in slab layout; dst = (buffer,
        bits, own_buffer, own_pattern): it is left there in slab layout (qsim_apply_ops_io: the re-layout's pack /
        unpack ride in the last / first fused pass).  parts (with dst): split form -- the slabs are stored piece by piece
        by `store_part(j)` for every piece of `pending_parts()`.  src_parts (with src): the source is still arriving in
        pieces: nothing runs until `load_part(j)` announces them, the first pass piece by piece.  tiles: the high tile bits
        of the first passes as the partition planner chose them (uint64 masks)."""
        st = self.chunk("state")
        if src is None and dst is None:
            if tiles is not None and len(tiles) and len(ops) >= 2:
                return st.apply_ops_tiled(ops, tiles)
            return st.apply_ops(ops)
        return st.apply_ops_io(ops, src=(self.chunk(src[0]), src[1]) if src else None,
                               dst=(self.chunk(dst[0]), dst[1], self.chunk(dst[2]), dst[3]) if dst else None, parts=parts,
                               src_parts=src_parts, tiles=tiles)

    def own_slab_in_state(self) -> bool:
        """The last `apply_ops` with a `dst` whose own-slab buffer was the source buffer left that slab in "state" (one
        pass read the source and stored the slabs: qsim_apply_ops_io_own_slab)."""
        return self.chunk("state").own_slab_in_chunk()

    def swap_names(self, a: str, b: str) -> None:
        """Buffers `a` and `b` trade names (the shard's home moves: the engine addresses buffers by role)."""
        self.tensor(a), self.tensor(b)
        self._tensors[a], self._tensors[b] = self._tensors[b], self._tensors[a]
        self._chunks[a], self._chunks[b] = self._chunks[b], self._chunks[a]

    def load_part(self, j: int) -> None:
        self.chunk("state").load_part(j)

    def pending_parts(self) -> list:
        return self.chunk("state").pending_parts()

    def store_part(self, j: int) -> None:
        self.chunk("state").store_part(j)

    # ---- exchange through the library's own communicator (DistributedEngine(exchange="cabi")) -----------------
    def comm_init(self, dist, rank: int, world: int) -> None:
        """qsim_comm over RCCL: rank 0 makes the unique id, the existing process group hands the 128 bytes around."""
        from quantum_simulations_amd.kernel.device import Comm
        box = [Comm.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        self.comm = Comm(self.device, rank, world, box[0])

    def exchange_bg(self, send: str, recv: str, entries) -> None:
        """entries [(peer, offset_amps, count_amps)] (one count): one RCCL group on the communicator's transfer stream,
        behind everything queued on the shard's stream so far; later work on that stream does not wait for it."""
        peers = [e[0] for e in entries]
        offs = [e[1] for e in entries]
        return self.comm.exchange_bg(peers, self.chunk(send), offs, self.chunk(recv), offs, entries[0][2])

    def exchange_wait(self, ticket: int) -> None:
        self.comm.wait(self.chunk("state"), ticket)

    def pack_all(self, bits, dst: str, skip_pattern: int, piece: int = 0, n_pieces: int = 1) -> None:
        self.chunk("state").pack_all(bits, self.chunk(dst), skip_pattern, piece, n_pieces)

    def unpack_all(self, bits, src: str, skip_pattern: int, piece: int = 0, n_pieces: int = 1) -> None:
        self.chunk("state").unpack_all(bits, self.chunk(src), skip_pattern, piece, n_pieces)

    def closed_form_error(self, kind: str, n_total: int, base_index: int, log_to_phys) -> float:
        return self.chunk("state").max_abs_err_closed_form(kind, n_total, base_index, log_to_phys)

    def fingerprint(self, n_total: int, base_index: int, log_to_phys, seed: int, sel_mask: int = 0, sel_value: int = 0) -> complex:
        return self.chunk("state").fingerprint(n_total, base_index, log_to_phys, seed, sel_mask, sel_value)

    def release_buffers(self) -> None:
        """Give the exchange buffers back to the device (they are re-created on demand): room for a one-GPU reference
        run of the whole state next to the shard."""
        self.sync()
        for name in [n for n in self._tensors if n != "state"]:
            self._chunks.pop(name).close()
            del self._tensors[name]
        self.torch.cuda.empty_cache()

    def profile_begin(self) -> None:
        self.chunk("state").profile_begin()

    def profile_end(self):
        return self.chunk("state").profile_end()

    def close(self) -> None:
        self.sync()
        if getattr(self, "comm", None) is not None:
            self.comm.close()
            self.comm = None
        for c in self._chunks.values():
            c.close()
        self._chunks.clear()
        self._tensors.clear()


def split_pieces(k: int, m: int, parts: int) -> list:
    """[(offset, amplitudes)] of the pieces the split form of qsim_apply_ops_io cuts every slab into -- the library's rule
    (qsim_split_piece_count: as many as asked for while a piece keeps >= 2^20 amplitudes; negative `parts`: no floor)
    restated for backends without the library (dry runs, the CPU test double; tests compare the two)."""
    want, floor = abs(parts), (20 if parts > 0 else 3)
    nb = 0
    while nb < 3 and (2 << nb) <= want and (k - m) - (nb + 1) >= floor:
        nb += 1
    piece = (1 << (k - m)) >> nb
    return [(j * piece, piece) for j in range(1 << nb)]


class Plan:
    """Step lists for successive executions (the staging layout carries over between them).
    `start_mappings[i]` is the planned layout execution i starts from; `execute` refuses a plan whose
    next execution was planned for another layout than the engine's current one."""

    def __init__(self, executions: list, mappings: list, start_mappings: list):
        self.executions, self.mappings, self.start_mappings, self.cursor = executions, mappings, start_mappings, 0


class _FakeTensor:
    """Stand-in for a shard / exchange buffer in dry runs: knows its length, checks slice bounds."""
    is_cuda = False

    def __init__(self, n: int):
        self.n = n

    def __getitem__(self, sl):
        start, stop, step = sl.indices(self.n) if isinstance(sl, slice) else (sl, sl + 1, 1)
        if not isinstance(sl, slice) or step != 1 or (sl.start or 0) < 0 or (sl.stop is not None and sl.stop > self.n) or stop < start:
            raise IndexError(f"slice {sl} outside a buffer of {self.n} elements")
        return _FakeTensor(stop - start)

    def numel(self) -> int:
        return self.n

    @staticmethod
    def element_size() -> int:
        return 8


class DryBackend:
    """No memory, no arithmetic: lets the engine run its communication schedule at full problem sizes
    (bench.py --dry-run, tests); every transfer is recorded in DistributedEngine.trace instead of posted."""
    dry = True

    def __init__(self, k: int):
        self.k = k
        self.local_passes = 0

    def tensor(self, name: str):
        return _FakeTensor(2 << self.k)

    def init_zero(self, set_amp0: bool) -> None:
        pass

    def sync(self) -> None:
        pass

    def apply_ops(self, ops, src=None, dst=None, parts: int = 0, src_parts: int = 0, tiles=None) -> int:
        self._loads = len(split_pieces(self.k, len(src[1]), src_parts)) if (src is not None and src_parts) else 0
        for side in (src, dst):
            if side is not None:
                self._check(side[1], 0, 1)
        self.local_passes += 1
        if dst is not None and parts:
            self._parts = split_pieces(self.k, len(dst[1]), parts)
        # (a dry run knows no pass counts: it takes the one-pass branch -- own slab into "state", buffers trade names --
        # whenever the engine offers it, which exercises the role bookkeeping; the transfers are the same either way)
        self._own_in_state = src is not None and dst is not None and dst[2] == src[0]
        return 1

    def own_slab_in_state(self) -> bool:
        return self._own_in_state

    def swap_names(self, a: str, b: str) -> None:
        pass

    def pending_parts(self) -> list:
        return self._parts

    def store_part(self, j: int) -> None:
        if not 0 <= j < len(self._parts):
            raise ValueError("bad part")

    def load_part(self, j: int) -> None:
        if not 0 <= j < self._loads:
            raise ValueError("bad source piece")

    def pack_all(self, bits, dst, skip_pattern, piece=0, n_pieces=1) -> None:
        self._check(bits, piece, n_pieces)

    def unpack_all(self, bits, src, skip_pattern, piece=0, n_pieces=1) -> None:
        self._check(bits, piece, n_pieces)

    def _check(self, bits, piece, n_pieces) -> None:
        if not 1 <= len(bits) <= 3 or len(set(bits)) != len(bits) or any(not 0 <= b < self.k for b in bits):
            raise ValueError(f"re-layout bits {bits} invalid for {self.k} local qubits")
        if n_pieces not in (1, 2, 4, 8) or not 0 <= piece < n_pieces:
            raise ValueError("bad piece")

    def close(self) -> None:
        pass


class PlanningBackend(DryBackend):
    """A dry backend that knows what the library WOULD do with every op list: the HBM passes of `qsim_apply_ops_io` (the host
    planner `qsim_plan_ops` on exactly the ops a rank runs, plus the pack / unpack passes of ends that cannot ride in a tile
    pass: slab bits inside a 128-byte line, a slab bit among the tile bits of the last pass, nothing to plan) and, from 26
    local qubits on, each pass weighted by the tile-cost model's prediction for its tile's index bits (runner/tile_layout.py,
    in units of the model's average pass).  `DistributedEngine.choose_initial_layout` executes candidate schedules on it."""

    def __init__(self, k: int):
        super().__init__(k)
        from quantum_simulations_amd.runner import tile_layout
        self._tile_layout = tile_layout
        self.model = tile_layout.model_for(k) if k >= 26 else None
        self.model_ref = 1.0
        if self.model is not None:
            rng = np.random.default_rng(7)
            top = min(k - 1, self.model["top"])
            self.model_ref = float(np.mean([tile_layout.tile_cost(self.model, rng.choice(np.arange(3, top + 1), size=8, replace=False))
                                            for _ in range(256)]))
        self._images = np.zeros((64, 4096), dtype=np.uint8)
        self.weight = 0.0                # model-weighted passes since the last reset
        self.passes = 0
        self.record: list | None = None  # (tools/shard_compute_probe.py: (op list, named tiles) as the rank would run them)

    def _plan(self, ops, tiles=None) -> tuple:
        """(passes, their model weight, tile bits of the last pass) of the fused plan of `ops` (tile passes possible)"""
        import ctypes as C

        from quantum_simulations_amd import _lib
        from quantum_simulations_amd.kernel.device import pack_ops
        nq, qubits, mats = pack_ops(ops)
        lib = _lib.load()
        count = C.c_int32()
        tm = np.ascontiguousarray(tiles if tiles is not None else [], dtype=np.uint64)
        args = (self.k, len(nq), nq.ctypes.data_as(C.c_void_p), qubits.ctypes.data_as(C.c_void_p), mats.ctypes.data_as(C.c_void_p),
                len(tm), tm.ctypes.data_as(C.c_void_p) if len(tm) else None)
        images = self._images
        _lib.check(lib.qsim_plan_ops_tiled(*args, images.ctypes.data_as(C.c_void_p), images.nbytes, C.byref(count)))
        if count.value > len(images):    # (a buffer too small only reports the count)
            images = self._images = np.zeros((2 * count.value, 4096), dtype=np.uint8)
            _lib.check(lib.qsim_plan_ops_tiled(*args, images.ctypes.data_as(C.c_void_p), images.nbytes, C.byref(count)))
        weight, last = 0.0, set()
        for p in range(count.value):     # (pass image = the kernel-argument block: T at byte 12, the tile's high bits from 16)
            T = int(images[p, 12:16].view("<i4")[0])
            last = {int(b) for b in images[p, 16:16 + T - 3]}
            weight += self._tile_layout.tile_cost(self.model, sorted(last)) / self.model_ref if self.model is not None else 1.0
        return count.value, weight, last

    def apply_ops(self, ops, src=None, dst=None, parts: int = 0, src_parts: int = 0, tiles=None) -> int:
        self._loads = len(split_pieces(self.k, len(src[1]), src_parts)) if (src is not None and src_parts) else 0
        for side in (src, dst):
            if side is not None:
                self._check(side[1], 0, 1)
        if dst is not None and parts:
            self._parts = split_pieces(self.k, len(dst[1]), parts)
        ops = list(ops)
        if self.record is not None and ops:
            self.record.append((ops, None if tiles is None else [int(m) for m in tiles]))
        if ops and 8 <= self.k <= 35:
            passes, weight, last = self._plan(ops, tiles)
        else:                            # (shards too small for tile passes: one launch per gate)
            passes, weight, last = len(ops), float(len(ops)), None
        tiles = last is not None and passes > 0
        fused_in = src is not None and tiles and min(src[1]) >= 3
        fused_out = dst is not None and tiles and min(dst[1]) >= 3 and not (set(dst[1]) & last)
        extra = int(src is not None and not fused_in) + int(dst is not None and not fused_out)
        self.last_extra = extra
        # (qsim_apply_ops_io_own_slab: ONE pass reads the source and stores the slabs)
        self._own_in_state = bool(src is not None and dst is not None and dst[2] == src[0] and fused_in and fused_out and passes == 1)
        self.local_passes += 1
        self.passes += passes + extra
        self.weight += weight + extra
        return passes + extra

    def pack_all(self, bits, dst, skip_pattern, piece=0, n_pieces=1) -> None:
        super().pack_all(bits, dst, skip_pattern, piece, n_pieces)
        self.weight += 1.0 / n_pieces

    def unpack_all(self, bits, src, skip_pattern, piece=0, n_pieces=1) -> None:
        super().unpack_all(bits, src, skip_pattern, piece, n_pieces)
        self.weight += 1.0 / n_pieces


class DistributedEngine:
    def __init__(self, n_qubits: int, world: int, rank: int, local_rank: int = 0,
                 mode: str = "fused", backend=None, staging: bool = True,
                 staging_method: str = "tiles", init_process_group: bool = True,
                 relayout_pieces: int = 4, min_piece_qubits: int = 20, fuse_relayout: bool = True,
                 rehearsal: bool = False, exchange: str = "torch", layout: str = "auto", pipeline_relayout: bool = True):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        if world & (world - 1) or world < 1:
            raise ValueError("world size must be a power of two")      # (1: no global qubit, nothing is ever exchanged: plumbing tests)
        self.n, self.world, self.rank = n_qubits, world, rank
        self.p = world.bit_length() - 1
        self.k = n_qubits - self.p
        if self.k < 2:
            # a dense gate on global qubits is brought local by trading places with local qubits (swap-and-stay): a
            # 2-qubit gate needs two local places; shards of fewer than 4 amplitudes are not supported
            raise ValueError(f"{n_qubits} qubits on {world} ranks leave {max(self.k, 0)} local qubit(s): at least 2 are needed "
                             "(use fewer ranks)")
        self.mode, self.staging, self.staging_method = mode, staging, staging_method
        self.tiles_min_ops = 0       # staging method "tiles": 0 = search the thin-pass threshold (plan_partition_best)
        self.plan_threads = 8
        self._plan_effort_high = False
        self.place_slots = True      # staging method "tiles", fresh state: local slots placed by the tile-cost model
        self.place_slots_min_k = 26  # (the model was fitted at 28 / 30 qubits; smaller shards are cache resident: tests lower it)
        self.use_tile_hints = True
        # rehearsal (explicit argument; bench.py --rehearsal): several ranks share the visible GPU(s), each with its shard
        # in HBM and the real HIP kernels, and exchange through host-staged gloo -- RCCL refuses two ranks on one
        # device.  `self.exchange` names what carries the transfers and is printed in every bench line.
        rehearsal = bool(rehearsal) and backend is None
        if backend is None:
            if rehearsal:   # several ranks share the visible GPU(s); exchange is host-staged over gloo
                local_rank = local_rank % max(1, torch.cuda.device_count())
            torch.cuda.set_device(local_rank)     # before RCCL initialises: one rank <-> one GPU
        if init_process_group and not dist.is_initialized():
            if backend is None and not rehearsal:
                dist.init_process_group("nccl", rank=rank, world_size=world,
                                        device_id=torch.device(f"cuda:{local_rank}"))
            else:
                dist.init_process_group("gloo", rank=rank, world_size=world)
        if exchange not in ("torch", "cabi"):
            raise ValueError("exchange must be 'torch' (torch.distributed P2P) or 'cabi' (qsim_comm_exchange)")
        self.exchange_api = exchange
        self.exchange = "gloo-rehearsal" if rehearsal else ("rccl" if backend is None else f"gloo ({type(backend).__name__})")
        self.backend = backend if backend is not None else HipShardBackend(self.k, local_rank)
        if exchange == "cabi":
            # the transfers are posted by the library's own RCCL communicator (qsim_comm_exchange_bg on its transfer
            # stream) instead of torch.distributed P2P: the same schedule, the path a C-only host uses
            if rehearsal or not hasattr(self.backend, "comm_init"):
                raise ValueError("exchange='cabi' needs one rank per GPU and the HIP backend (RCCL inside libqsim_hip.so)")
            self.backend.comm_init(dist, rank, world)
        self.dry = bool(getattr(self.backend, "dry", False))
        self.trace: list | None = [] if self.dry else None     # dry runs: (kind, peer, sent, received) per posted transfer
        self.trace_posts = 0                                   # dry runs: groups posted (>= 2 per fused re-layout: pieces)
        self.l2p_planned = list(range(n_qubits))   # logical qubit -> physical index bit as the PLANS see it
        self._dyn = list(range(n_qubits))          # planned physical bit -> actual physical bit (swap-and-stay moves)
        self._flat: list = []                      # qubit lists of the running execution, in order (victim choice)
        self._flat_pos = 0
        self.xgmi_bytes_sent = 0
        self.exchanges = 0
        self._comm_events: list = []
        if relayout_pieces not in (1, 2, 4, 8):
            raise ValueError("relayout_pieces must be 1, 2, 4 or 8")
        self.relayout_pieces, self.min_piece_qubits = relayout_pieces, min_piece_qubits
        # Initial qubit layout (round 4): |0..0> is the same state under every assignment of qubits to index bits, so the first
        # plan after `init_zero_state` may start from any -- which qubits are global first, which three sit on the line bits
        # (they belong to every tile) -- and the staged schedule that follows differs in re-layouts and HBM passes:
        # `plan` tries LAYOUT_CANDIDATES random assignments beside the identity and keeps the cheapest (`_candidate_cost`: each one executed on a planning twin of this engine).
        # "auto": shards of >= 20 local qubits (staged or swap-and-stay schedules alike); "search": always (tests); "identity": never.
        if layout not in ("auto", "search", "identity"):
            raise ValueError("layout must be 'auto', 'search' or 'identity'")
        self.layout = layout
        self._fresh = False                        # the state is |0..0> and no plan has chosen a layout for it yet
        self._shadow = None                        # (the planning twin that prices candidate layouts, made on first use)
        self.relayout_log: list = []               # slab-bit counts of the re-layouts made since the last reset (shadow engines)
        self.layout_info = None
        self._passes = self.last_passes = 0
        self.home_moves = 0                        # times "state" and "buf1" traded names (one-pass op list between two re-layouts)
        self._pending: list = []
        self._pending_tiles: list = []             # tile masks the planner named for the passes of the queued ops
        # Re-layout fused with the neighbouring local passes (round 3): the last fused pass before an exchange stores
        # its tiles straight into the send buffer in slab order and the first one after it loads them from the
        # receive buffer -- no separate pack / unpack pass of the shard.  `_state_in` = (buffer, local bits) while
        # the shard lives in a receive buffer in slab layout (None: in "state", index order).
        self.fuse_relayout = fuse_relayout
        # pipeline_relayout = False (bench.py --no-relayout-pipeline): the plain form of a fused re-layout -- whole slabs
        # stored by one launch, ONE group posted, and the host path WAITS for it before anything reads the shard (no pieces,
        # nothing `_inflight`).  The in-flight piece pipeline has only ever run on one GPU and under gloo rehearsal
        # (ADVICE r04): a first multi-GPU run that fails can be repeated with this switch to tell a wrong schedule from an
        # ordering problem between the transfer stream and the partial launches.
        self.pipeline_relayout = bool(pipeline_relayout)
        self._state_in = None
        # ... and, while the pieces of that re-layout may still be on the links, `_inflight` = (posted groups, timer): the
        # next reader of the shard consumes them piece by piece (its first pass starts on the tiles whose pieces are there)
        self._inflight = None
        # Memory per rank: the shard + the send buffer + the receive buffer = 3 shard-sized allocations (48 GiB at 30 local
        # qubits, 192 GiB at 32), fused re-layouts or not: a pass that reads the shard from the receive buffer and stores
        # the next re-layout's slabs leaves its own slab in "state" (free at that time) and the two buffers trade names
        # (`relayout`); round 3 kept a second receive buffer for that.  Claimed here, not in the middle of a circuit.
        if not self.dry and hasattr(self.backend, "tensor"):
            try:
                for name in ("buf0", "buf1"):
                    self.backend.tensor(name)
            except RuntimeError as e:        # torch's out-of-memory error is a RuntimeError
                raise MemoryError(f"rank {rank}: no room for the exchange buffers (3 x {16 << self.k} bytes per rank are "
                                  f"needed): {e}") from e

    # ---- layout ------------------------------------------------------------------------
    @property
    def l2p(self) -> list[int]:
        """logical qubit -> ACTUAL physical index bit (planned layout composed with the dynamic moves)"""
        return [self._dyn[p] for p in self.l2p_planned]

    # ---- helpers -----------------------------------------------------------------------
    def _rank_bit(self, phys_qubit: int) -> int:
        return (self.rank >> (phys_qubit - self.k)) & 1

    def _post(self, send: str, recv: str, entries):
        """Post entries [(peer, start, count)] together -- `count` float64 elements at offset `start` of buffer `send` go
        to `peer`, as many arrive from it at the same place of `recv` -- without waiting (RCCL: the group is ordered after
        everything already queued on the shard's stream and runs beside what is queued later).  One group = every peer
        at once: all links busy."""
        dist, torch = self.dist, self.torch
        if self.dry:
            self.trace_posts += 1
            for peer, start, count in entries:
                self.backend.tensor(send)[start:start + count]          # (slice bounds are checked)
                self.backend.tensor(recv)[start:start + count]
                self.trace.append((self._trace_kind, int(peer), count * 8, count * 8))
                self.xgmi_bytes_sent += count * 8
            return ([], [])
        if not entries:
            return ([], [])
        self.xgmi_bytes_sent += sum(c for _, _, c in entries) * 8
        if self.exchange_api == "cabi":
            ticket = self.backend.exchange_bg(send, recv, [(peer, start // 2, count // 2) for peer, start, count in entries])
            return ("cabi", ticket)
        st, rt = self.backend.tensor(send), self.backend.tensor(recv)
        staged, ops = [], []
        host = dist.get_backend() == "gloo"
        for peer, start, count in entries:
            s_t, r_t = st[start:start + count], rt[start:start + count]
            if host and s_t.is_cuda:   # rehearsal of several ranks on one GPU: stage through host
                s_h, r_h = s_t.cpu(), torch.empty(r_t.shape, dtype=r_t.dtype)
                staged.append((r_t, r_h))
                s_t, r_t = s_h, r_h
            ops.append(dist.P2POp(dist.isend, s_t, peer))
            ops.append(dist.P2POp(dist.irecv, r_t, peer))
        return (dist.batch_isend_irecv(ops), staged)

    def _finish(self, posted) -> None:
        """Received data may be used by what is queued after this (RCCL: the shard's stream waits, not the host)."""
        works, staged = posted
        if works == "cabi":
            self.backend.exchange_wait(staged)
            return
        for work in works:
            work.wait()
        for dev_t, host_t in staged:
            dev_t.copy_(host_t)

    _trace_kind = "relayout"

    def _comm_timer(self, tensor):
        """Device-side time of an exchange (stream events, summed in comm_stats)."""
        if not (tensor.is_cuda and self.dist.get_backend() == "nccl"):
            return None
        ev0, ev1 = self.torch.cuda.Event(enable_timing=True), self.torch.cuda.Event(enable_timing=True)
        ev0.record()
        return ev0, ev1

    def _comm_done(self, timer) -> None:
        if timer is not None:
            timer[1].record()
            self._comm_events.append(timer)
        self.exchanges += 1

    # ---- deferred local work -------------------------------------------------------------------
    # Global-qubit gates that need no exchange end up as LOCAL ops on this rank (a rank-bit phase,
    # a diagonal on the local partner qubit, the conditional 1q gate of a global control).  Run one
    # by one each is a full HBM pass of the shard; queued, they ride in the next fused local pass
    # (program order is kept: they sit between two local batches), or are flushed before anything
    # that reads the shard (an exchange, a re-layout, a reduction, a download).
    def _queue_local(self, op) -> None:
        self._fresh = False
        self._pending.append(op)

    def _run_local(self, ops, dst=None, parts: int = 0) -> None:
        """`ops` (may be empty) on the shard wherever it lives -- in "state", in a receive buffer in slab layout
        (`_state_in`), or still arriving there piece by piece (`_inflight`): then the backend plans now and every piece
        is handed over as soon as its transfer is done (torch: the shard's stream waits for that group, not the host), so
        the first pass runs on the tiles whose pieces are there while the later pieces are on the links.  `dst` / `parts`:
        the slab-storing end of the next re-layout."""
        src = self._state_in
        kw = {}
        if self._pending_tiles:          # (the partition planner's tiles for this op list: staging method "tiles")
            kw["tiles"], self._pending_tiles = np.array(self._pending_tiles, dtype=np.uint64), []
        if src is None:
            if dst is None:
                if ops:
                    self._passes += self.backend.apply_ops(ops, **kw) or 0
            else:
                self._passes += self.backend.apply_ops(ops, dst=dst, parts=parts, **kw) or 0
            return
        self._state_in = None
        if self._inflight is None:
            self._passes += self.backend.apply_ops(ops, src=src, dst=dst, parts=parts, **kw) or 0
            return
        posted, timer = self._inflight
        self._inflight = None
        self._passes += self.backend.apply_ops(ops, src=src, dst=dst, parts=parts, src_parts=self._split_parts(), **kw) or 0
        for j, pst in enumerate(posted):
            self._finish(pst)
            self.backend.load_part(j)
        self._comm_done(timer)

    def _flush_local(self) -> None:
        """Run the queued local ops (reading the shard from the receive buffer it may still live in, or be arriving in);
        with nothing queued a shard that is not at home is brought there (unpack pieces)."""
        if self._pending or self._state_in is not None:
            ops, self._pending = self._pending, []
            self._run_local(ops)

    def _drain_inflight(self) -> None:
        """The state is about to be overwritten: wait for transfers that still write into the exchange buffers."""
        if self._inflight is not None:
            posted, timer = self._inflight
            self._inflight = None
            for pst in posted:
                self._finish(pst)
            self._comm_done(timer)

    # ---- state ---------------------------------------------------------------------------
    def init_zero_state(self) -> None:
        self._drain_inflight()
        self._pending = []
        self._pending_tiles = []
        self._state_in = None
        self.backend.init_zero(self.rank == 0)
        self.l2p_planned = list(range(self.n))
        self._dyn = list(range(self.n))
        self._fresh = True
        self.layout_info = None

    def norm2(self) -> float:
        self._flush_local()
        t = self.torch.tensor([self.backend.norm2()], dtype=self.torch.float64)
        if self.dist.get_backend() == "nccl":
            t = t.cuda()
        self.dist.all_reduce(t)
        return float(t.item())

    def state_vector(self) -> np.ndarray:
        """Whole state in LOGICAL qubit order on every rank (small n only: tests, examples)."""
        self._flush_local()
        local = self.torch.from_numpy(self.backend.download().view(np.float64).copy())
        parts = [self.torch.empty_like(local) for _ in range(self.world)]
        if self.dist.get_backend() == "nccl":
            dev = self.backend.tensor("state").device
            local, parts = local.to(dev), [p.to(dev) for p in parts]
        self.dist.all_gather(parts, local)
        full = np.concatenate([p.cpu().numpy().view(np.complex128) for p in parts])
        return permute_state(full, self.l2p)

    # ---- planning --------------------------------------------------------------------------
    def _tiles_method(self) -> bool:
        """Staging method "tiles" applies: shards large enough for tile passes (the planner is the library's pass builder)."""
        return bool(self.staging and self.staging_method == "tiles" and 8 <= self.k <= 35 and self.n <= 63)

    def _fused_ops(self, cd: dict, l2p: list) -> list:
        """The circuit as an op list on the index bits of layout `l2p`, runs of 1q gates fused (fusion.py:41-81)."""
        from quantum_simulations_amd.circuit.fusion import fuse_1q_ops
        return fuse_1q_ops([([l2p[q] for q in g["qubits"]], gate_table.gate_matrix(g["gate"], g["params"])) for g in cd["gates"]])

    def _packed_ops(self, cd: dict):
        """`_fused_ops` in logical labels, packed once per circuit for the partition planner (relabelled per layout)."""
        from quantum_simulations_amd.runner.partition_plan import PackedOps
        key = id(cd)
        if getattr(self, "_packed_key", None) != key:
            self._packed, self._packed_key, self._packed_cd = PackedOps(self._fused_ops(cd, list(range(self.n))), self.n), key, cd
        return self._packed

    def _steps_from(self, cd: dict, l2p: list[int]):
        """Plan `cd` for a state whose logical qubit q currently sits at physical bit l2p[q]."""
        relabeled = {"number_of_qubits": self.n,
                     "gates": [{"qubits": [l2p[q] for q in g["qubits"]], "gate": g["gate"],
                                "params": g["params"]} for g in cd["gates"]]}
        if self._tiles_method():
            # stage boundaries and tile passes planned together (runner/partition_plan.py)
            from quantum_simulations_amd.runner.partition_plan import MIN_OPS_CHOICES, plan_partition, plan_partition_best
            ops = self._packed_ops(cd).relabeled(l2p)
            if self.tiles_min_ops:
                res = plan_partition(ops, self.n, self.k, min_ops=self.tiles_min_ops, relayout_cost=self.RELAYOUT_PASSES)
            else:
                # (plans that run a few times search less: planning is host time the caller waits for)
                choices = MIN_OPS_CHOICES if self._plan_effort_high else (16, 24)
                res = plan_partition_best(ops, self.n, self.k, choices=choices, relayout_cost=self.RELAYOUT_PASSES, threads=self.plan_threads)
            self.last_partition_plan = res
            steps, moved = res["steps"], res["moved"]
        elif self.staging and self.k >= 2:   # (staging cannot hold a 2-qubit gate in fewer than 2 local qubits)
            # ("tiles" on shards too small for tile passes: the stage-by-stage method it replaces)
            steps, moved = atlas_stages(relabeled, self.k, method="belady" if self.staging_method == "tiles" else self.staging_method,
                                        strict_order=True)
        else:
            steps, moved = batch_levels(levelize(relabeled), self.k), list(range(self.n))
        return steps, [moved[l2p[q]] for q in range(self.n)]

    # ---- the initial layout ---------------------------------------------------------------------
    LAYOUT_CANDIDATES = 48
    # An all-to-all over m bits in units of one fused pass of the shard: (2^-m of the shard to each of 2^m - 1 peers, each
    # over its own xGMI link at 0.8 x 153 GB/s) / (the shard read and written once at 5 TB/s).  A MODEL -- no multi-GPU
    # node was available to measure it -- used only to weigh re-layouts against passes when two layouts differ in both.
    RELAYOUT_PASSES = {1: 10.4, 2: 5.2, 3: 2.6}

    def _candidate_cost(self, cd: dict, l2p: list, repeats: int = 1) -> tuple:
        """(cost in pass units per execution, HBM passes of the first execution, its re-layout sizes) of executing `cd`
        `repeats` times from |0..0> in the layout `l2p` ON THIS RANK: a shadow engine with a `PlanningBackend` runs the real
        schedule code -- staging, deferred local batches, rank-bit phases and conditional gates of this rank, fused
        re-layout ends -- without memory or arithmetic.  Later executions start from the layout the one before left behind;
        at most three are run, the mean of the second and third standing for all later ones."""
        sh = self._shadow
        if sh is None:
            sh = self._shadow = DistributedEngine(self.n, self.world, self.rank, mode=self.mode, backend=PlanningBackend(self.k),
                                                  staging=self.staging, staging_method=self.staging_method, init_process_group=False,
                                                  relayout_pieces=self.relayout_pieces, min_piece_qubits=self.min_piece_qubits,
                                                  fuse_relayout=self.fuse_relayout, layout="identity")
        sh.staging = self.staging
        sh.init_zero_state()
        sh._fresh = False
        sh.l2p_planned = list(l2p)
        run = max(1, min(repeats, 3))
        plan = sh.plan(cd, repeats=run)
        costs, first = [], None
        for _ in range(run):
            sh.backend.weight, sh.backend.passes, sh.relayout_log = 0.0, 0, []
            sh.execute(plan)
            costs.append(sh.backend.weight + sum(self.RELAYOUT_PASSES[m] for m in sh.relayout_log))
            first = first or (sh.last_passes, list(sh.relayout_log))
        later = float(np.mean(costs[1:])) if run > 1 else 0.0
        return (costs[0] + (max(1, repeats) - 1) * later) / max(1, repeats), first[0], first[1]

    def choose_initial_layout(self, cd: dict, n_candidates: int | None = None, seed: int = 20260504, repeats: int = 1) -> list:
        """l2p for a state that is still |0..0>: the identity or one of `n_candidates` random assignments, whichever gives
        the staged schedule of `cd` the lowest cost on the SLOWEST rank (`_candidate_cost` per rank, maximum over the ranks:
        they run different op lists; ties: the earlier candidate, the identity first).  COLLECTIVE."""
        n_candidates = self.LAYOUT_CANDIDATES if n_candidates is None else n_candidates
        rng = np.random.default_rng(seed)
        cands = [list(range(self.n))] + [[int(x) for x in rng.permutation(self.n)] for _ in range(n_candidates)]
        scored = [self._candidate_cost(cd, l2p, repeats) for l2p in cands]
        costs = self.torch.tensor([c for c, _, _ in scored], dtype=self.torch.float64)
        if self.dist.is_initialized() and self.world > 1:
            if self.dist.get_backend() == "nccl":
                costs = costs.cuda()
            self.dist.all_reduce(costs, op=self.dist.ReduceOp.MAX)
            costs = costs.cpu()
        best = min(range(len(cands)), key=lambda i: (float(costs[i]), i))
        self.layout_info = {"candidates": len(cands), "executions_planned_for": max(1, repeats),
                            "identity": {"cost_max_over_ranks": round(float(costs[0]), 2), "passes_this_rank": scored[0][1], "relayouts": scored[0][2]},
                            "chosen": {"cost_max_over_ranks": round(float(costs[best]), 2), "passes_this_rank": scored[best][1],
                                       "relayouts": scored[best][2], "index": best}}
        return cands[best]

    LAYOUT_MIN_REPEATS = 8          # layout "auto": plans for fewer executions try 2 start layouts, not 17 (the search is host time)

    def choose_initial_layout_tiles(self, cd: dict, repeats: int = 1, n_candidates: int | None = None, seed: int = 20260504) -> list:
        """Staging method "tiles": l2p for a state that is still |0..0>.  Candidates: the identity, and assignments that put
        the p qubits whose FIRST use as a target comes last on the rank bits (Belady at time zero) with the other qubits
        in random order (which three sit on the line bits, members of every tile, moves the pass count), one in eight any
        assignment at all.  Each is priced by
        the partition planner itself -- passes + re-layouts in pass units of the first execution, and of a second one from
        the layout the first leaves behind when the plan will be repeated -- in parallel threads.  The planner names its
        tiles to the library, so what is priced is what every rank runs: no twin execution, no collective.  Deterministic."""
        import time

        from quantum_simulations_amd.runner.partition_plan import plan_partition, planning_pool
        t0 = time.perf_counter()
        n, k, p = self.n, self.k, self.p
        if n_candidates is None:
            n_candidates = 16 if self._plan_effort_high else 1     # (32 found nothing better on the seeded workloads)
        packed = self._packed_ops(cd)
        first = [1 << 60] * n
        for i, tg in enumerate(packed.targets):
            for q in tg:
                first[q] = min(first[q], i)
        far = sorted(range(n), key=lambda q: (-first[q], -q))[:p]
        rng = np.random.default_rng(seed)
        cands = [list(range(n))]
        for c in range(n_candidates):
            if c and c % 8 == 7:                      # (one in eight: any assignment at all)
                cands.append([int(x) for x in rng.permutation(n)])
                continue
            rest = [int(q) for q in (rng.permutation(n) if c else np.arange(n)) if q not in far]
            l2p = [0] * n
            for i, q in enumerate(rest):
                l2p[q] = i
            for i, q in enumerate(sorted(far)):
                l2p[q] = k + i
            cands.append(l2p)

        def price(l2p):
            costs, first_exec = [], None
            for _ in range(2 if repeats > 1 else 1):
                r = plan_partition(packed.relabeled(l2p), n, k, min_ops=self.tiles_min_ops or 24, relayout_cost=self.RELAYOUT_PASSES)
                costs.append(r["cost"])
                first_exec = first_exec or (r["passes"], r["relayouts"])
                l2p = [r["moved"][l2p[q]] for q in range(n)]
            later = costs[-1]
            return (costs[0] + (max(1, repeats) - 1) * later) / max(1, repeats), first_exec[0], first_exec[1]
        scored = list(planning_pool(self.plan_threads).map(price, cands)) if self.plan_threads > 1 else [price(c) for c in cands]
        best = min(range(len(cands)), key=lambda i: (scored[i][0], i))
        self.layout_info = {"candidates": len(cands), "executions_planned_for": max(1, repeats), "method": "tiles",
                            "identity": {"cost_max_over_ranks": round(scored[0][0], 2), "passes_this_rank": scored[0][1], "relayouts": scored[0][2]},
                            "chosen": {"cost_max_over_ranks": round(scored[best][0], 2), "passes_this_rank": scored[best][1],
                                       "relayouts": scored[best][2], "index": best},
                            "search_seconds": round(time.perf_counter() - t0, 3)}
        return cands[best]

    def plan(self, circuit_dict: dict, repeats: int = 1, effort: str | None = None) -> Plan:
        """Step lists for `repeats` successive executions from the engine's current layout.  COLLECTIVE when it is the first
        plan of a freshly initialised state and the engine searches the initial layout (every rank must call it: rank 0's
        choice is broadcast); host-only otherwise."""
        cd = validate_circuit_dict(circuit_dict)
        if cd["number_of_qubits"] != self.n:
            raise ValueError(f"circuit has {cd['number_of_qubits']} qubits, engine has {self.n}")
        # effort: "high" = the full search of start layouts and thin-pass thresholds (seconds of host time: worth it for a
        # plan that runs many times), "low" = two start layouts, two thresholds; None: by `repeats`
        self._plan_effort_high = (effort == "high") if effort else (repeats >= self.LAYOUT_MIN_REPEATS or self.layout == "search")
        was_fresh = self._fresh and self.layout != "identity"
        if self._fresh:
            # (once per initialised state: a second plan made before the first one runs keeps this layout, so both stay valid)
            self._fresh = False
            if self.world > 1 and self.k >= 2 and (self.layout == "search" or (self.layout == "auto" and self.k >= 20)):
                if self._tiles_method():
                    self.l2p_planned = self.choose_initial_layout_tiles(cd, repeats=max(1, repeats))
                else:
                    import time
                    t0 = time.perf_counter()
                    self.l2p_planned = self.choose_initial_layout(cd, repeats=max(1, repeats))
                    self.layout_info["search_seconds"] = round(time.perf_counter() - t0, 3)
        executions, mappings, starts = [], [], []
        l2p = list(self.l2p_planned)
        for _ in range(max(1, repeats)):
            starts.append(list(l2p))
            steps, l2p = self._steps_from(cd, l2p)
            executions.append(steps)
            mappings.append(list(l2p))
        if was_fresh and self.place_slots and self._tiles_method() and self.k >= self.place_slots_min_k:
            # |0..0> looks the same under every assignment of qubits to index bits: the local slots of the whole chain of
            # executions are put on the index bits whose tiles have the best DRAM pattern (partition_plan.place_slots)
            import time

            from quantum_simulations_amd.runner.partition_plan import place_slots
            t0 = time.perf_counter()
            sigma, before, after = place_slots(executions, self.k)
            if sigma is not None:
                mp = lambda b: sigma.get(b, b)                           # noqa: E731
                starts = [[mp(b) for b in m] for m in starts]
                mappings = [[mp(b) for b in m] for m in mappings]
                self.l2p_planned = list(starts[0])
                self.layout_info = dict(self.layout_info or {}, slot_placement={
                    "tile_model_ms_per_plan": [round(before, 2), round(after, 2)], "seconds": round(time.perf_counter() - t0, 3)})
        return Plan(executions, mappings, starts)

    def passes_per_step(self, plan: Plan) -> int:
        """HBM passes of the last executed circuit on this rank: fused tile launches of the local steps, + 1 for every
        pack / unpack of a re-layout that could not ride in a neighbouring fused pass (2 per re-layout with
        fuse_relayout=False); before any execution, the op count of the plan."""
        if self.last_passes:
            return self.last_passes
        return sum(len(s["local_ops"]) + len(s["nonlocal_ops"]) for s in plan.executions[0])

    # ---- execution ---------------------------------------------------------------------------
    def execute(self, plan: Plan) -> None:
        i = plan.cursor
        if i >= len(plan.executions):
            raise RuntimeError("plan exhausted: call engine.plan(circuit, repeats=K) with enough repeats")
        if plan.start_mappings[i] != self.l2p_planned:
            raise RuntimeError("this execution of the plan was planned for another qubit layout than the engine's "
                               "current one (the state was re-initialised or another plan ran in between): re-plan")
        self._passes = 0
        self._flat = [qs for step in plan.executions[i] for qs, _ in list(step["local_ops"]) + list(step["nonlocal_ops"])]
        self._flat_pos = 0
        for step in plan.executions[i]:
            self.run_step(step)
        self._flush_local()
        self.last_passes = self._passes
        self.l2p_planned = list(plan.mappings[i])
        plan.cursor = i + 1

    def _actual(self, qs) -> list[int]:
        return [self._dyn[q] for q in qs]

    def run_step(self, step: dict) -> None:
        """Ops carry PLANNED physical bits; swap-and-stay moves may have put a planned-local qubit on a
        rank bit (and back), so every op is classified by where its qubits actually are."""
        k = self.k
        self._fresh = False              # (the state is no longer |0..0>: no later plan may pick another layout for it)
        if step.get("tile_masks") and self.use_tile_hints:
            self._pending_tiles += [int(m) for m in step["tile_masks"]]
        batch = []
        for qs, U in step["local_ops"]:
            aq = self._actual(qs)
            if all(q < k for q in aq):
                batch.append((aq, U))
            else:                                   # a victim of an earlier move: the ops before it first
                self._pending += batch
                batch = []
                self.apply_nonlocal(aq, U)
            self._flat_pos += 1
        self._pending += batch           # runs with the next flush: before an exchange (whose pack it then absorbs),
        ops = step["nonlocal_ops"]       # a reduction, a download, or at the end of the execution
        i = 0
        while i < len(ops):
            j = i
            group = []
            used: set[int] = set()
            while (j < len(ops) and len(group) < 3 and self._is_planned_swap(ops[j])
                   and self._is_cross(self._actual(ops[j][0])) and used.isdisjoint(self._actual(ops[j][0]))):
                group.append(self._actual(ops[j][0]))   # (at most 3 pairs: qsim_pack_all's slab patterns)
                used.update(group[-1])
                j += 1
            if group:
                self.relayout(group)
                self._flat_pos += j - i
                i = j
                continue
            qs, U = ops[i]
            if self._is_planned_swap(ops[i]):
                # a planned SWAP whose qubits are on the same side now: nothing has to move -- the two
                # planned positions trade their actual bits (later gates find the qubits where they are)
                a, b = qs
                self._dyn[a], self._dyn[b] = self._dyn[b], self._dyn[a]
            else:
                aq = self._actual(qs)
                if all(q < k for q in aq):
                    self._queue_local((aq, U))
                else:
                    self.apply_nonlocal(aq, U)
            self._flat_pos += 1
            i += 1

    def _is_cross(self, aq) -> bool:
        return (aq[0] < self.k) != (aq[1] < self.k)

    @staticmethod
    def _is_planned_swap(op) -> bool:
        qs, U = op
        return len(qs) == 2 and U.shape == (4, 4) and np.array_equal(U, _SWAP)

    # -- all-to-all re-layout: swap m local bits with m global bits ------------------------------
    def relayout(self, pairs) -> None:
        """pairs: [[p_a, p_b], ...] each with exactly one local and one global physical bit.

        Fused (the default, slab bits above the line bits): the queued local ops' last pass stores the slabs, one
        grouped exchange of whole slabs, the next local pass loads them -- no pass of the shard outside the links.
        Unfused: pack / exchange / unpack, pipelined in `pieces` sub-ranges of every slab (while piece s is on the
        links, piece s+1 is being packed and piece s-1 unpacked)."""
        self._fresh = False
        loc = [min(p) for p in pairs]
        glo = [max(p) for p in pairs]
        m = len(pairs)
        self.relayout_log.append(m)
        slab = 2 << (self.k - m)                       # float64 elements per slab
        mine = sum(((self.rank >> (g - self.k)) & 1) << i for i, g in enumerate(glo))
        send = self.backend.tensor("buf0")
        peers = []
        for d in range(1 << m):
            if d == mine:
                continue
            peer = self.rank
            for i, g in enumerate(glo):
                peer = (peer & ~(1 << (g - self.k))) | (((d >> i) & 1) << (g - self.k))
            peers.append((d, peer))
        pieces = self._relayout_pieces(self.k - m)
        part = slab // pieces
        if self.fuse_relayout and min(loc) >= 3 and self.k - m >= 3:
            # Fused: the queued local ops' last pass writes the slabs (own slab straight into the receive buffer), the next
            # local pass will read them from there.  (A slab bit inside a 128-byte line would break whole-line accesses:
            # the unfused path below handles it.)
            # What overlaps what: the slab-storing pass is cut into up to `relayout_pieces` PIECES
            # (the j-th equal sub-range of every slab: qsim_ops_io::dst_parts) and the exchange of piece j -- one group
            # with all 2^m - 1 peers, every link busy -- is posted as soon as piece j is stored, so it travels while the
            # pieces behind it are computed; only the first piece's compute and the last piece's transfer are exposed on
            # the send side.  The cut depends only on (k, m, pieces): all ranks post the same messages in the same order
            # whatever their own pass plans look like (a rank whose last pass holds a piece bit as a tile bit has all its
            # pieces ready at once: it overlaps less, it does not post differently).  The first pass AFTER the exchange
            # takes the pieces over as they arrive (`_inflight`, qsim_ops_io::src_parts): it runs on the tiles whose
            # pieces are there -- when the piece bits are no tile bits of it -- while the later pieces are on the links.
            # Buffers: the slabs go into "buf0", the slab that stays straight into the receive buffer "buf1" -- also when
            # the shard currently LIVES in "buf1" (two re-layouts with little between them): with two or more kernels "buf1"
            # has been consumed by the first before the last stores into it; when ONE pass reads "buf1" and stores the
            # slabs, the library leaves the own slab in "state" instead (nobody needs its contents then), the exchange
            # delivers into "state", and "state" and "buf1" trade names afterwards: the shard is in "buf1" again and the
            # consumed buffer is the new home.  Three shard-sized buffers per rank in every case.
            from_recv = self._state_in is not None and self._state_in[0] == "buf1"
            ops, self._pending = self._pending, []
            if not self.pipeline_relayout:
                # plain form: the slabs are stored by this call, one group carries them whole, and it is waited for
                self._drain_inflight()
                self._run_local(ops, dst=("buf0", loc, "buf1", mine), parts=0)
                rname = "state" if (from_recv and self.backend.own_slab_in_state()) else "buf1"
                timer = self._comm_timer(send)
                posted = self._post("buf0", rname, [(peer, d * slab, slab) for d, peer in peers])
                self._finish(posted)
                self._comm_done(timer)
                if rname == "state":
                    self.backend.swap_names("state", "buf1")
                    self.home_moves += 1
                self._state_in = ("buf1", list(loc))
                return
            self._run_local(ops, dst=("buf0", loc, "buf1", mine), parts=self._split_parts())
            rname = "state" if (from_recv and self.backend.own_slab_in_state()) else "buf1"
            timer = self._comm_timer(send)
            posted = []
            for j, (off, cnt) in enumerate(self.backend.pending_parts()):
                self.backend.store_part(j)
                posted.append(self._post("buf0", rname, [(peer, d * slab + 2 * off, 2 * cnt) for d, peer in peers]))
            if rname == "state":
                self.backend.swap_names("state", "buf1")       # (posted transfers hold the buffers themselves, not the names)
                self.home_moves += 1
            # nobody waits here: the next reader of the shard takes the pieces over as they arrive (_run_local)
            self._inflight = (posted, timer)
            self._state_in = ("buf1", list(loc))
            return
        self._flush_local()
        self._passes += 2
        timer = self._comm_timer(send)
        posted = []
        self.backend.pack_all(loc, "buf0", mine, 0, pieces)        # slab d at offset d * 2^(k-m)
        for s in range(pieces):
            posted.append(self._post("buf0", "buf1", [(peer, d * slab + s * part, part) for d, peer in peers]))
            if s + 1 < pieces:
                self.backend.pack_all(loc, "buf0", mine, s + 1, pieces)
        for s in range(pieces):
            self._finish(posted[s])
            self.backend.unpack_all(loc, "buf1", mine, s, pieces)
        self._comm_done(timer)

    def _split_parts(self) -> int:
        """qsim_ops_io::dst_parts / src_parts of a fused re-layout: the configured pieces (the library keeps a piece >=
        2^20 amplitudes); negative (no floor) when the engine was built with a lower floor (tests on small shards)."""
        want = self.relayout_pieces
        return want if self.min_piece_qubits >= 20 else -want

    def _relayout_pieces(self, slab_qubits: int) -> int:
        """Pieces per slab: the configured count, capped so that a piece keeps >= 2^20 amplitudes
        (16 MiB per peer and piece -- large enough for full link rate)."""
        want = self.relayout_pieces
        while want > 1 and slab_qubits - (want.bit_length() - 1) < self.min_piece_qubits:
            want //= 2
        return max(1, want)

    # -- one gate with at least one global qubit ---------------------------------------------------
    def apply_nonlocal(self, qs, U) -> None:
        """qs: ACTUAL physical bits, at least one >= k."""
        k = self.k
        self._fresh = False
        if len(qs) == 1:
            q = qs[0]
            b = self._rank_bit(q)
            if _is_diagonal(U):                       # rank-bit phase, no exchange
                if U[b, b] != 1:
                    self._scale(U[b, b])
                return
            (v,) = self._bring_local([q], exclude=())
            self._queue_local(([v], U))
            return
        qa, qb = qs
        a_glob, b_glob = qa >= k, qb >= k
        if _is_diagonal(U):                           # CZ / CR / any diagonal: no exchange
            d = np.diag(U).reshape(2, 2)              # d[bit a][bit b]
            if a_glob and b_glob:
                f = d[self._rank_bit(qa), self._rank_bit(qb)]
                if f != 1:
                    self._scale(f)
            elif a_glob:
                row = d[self._rank_bit(qa)]
                if not (row[0] == 1 and row[1] == 1):
                    self._queue_local(([qb], np.diag(row)))
            else:
                col = d[:, self._rank_bit(qb)]
                if not (col[0] == 1 and col[1] == 1):
                    self._queue_local(([qa], np.diag(col)))
            return
        V = _controlled_on_first(U)
        ctrl, tgt = qa, qb
        if V is None:
            V = _controlled_on_second(U)
            ctrl, tgt = qb, qa
        if V is not None and ctrl >= k:               # global control: conditional 1q gate
            if tgt < k:
                if self._rank_bit(ctrl):
                    self._queue_local(([tgt], V))
            elif _is_diagonal(V):
                if self._rank_bit(ctrl):
                    self.apply_nonlocal([tgt], V)
            else:
                # the target has to come local on EVERY rank (the move is collective); the gate keeps its
                # global control and is applied by the ranks whose control bit is 1
                (v,) = self._bring_local([tgt], exclude=())
                if self._rank_bit(ctrl):
                    self._queue_local(([v], V))
            return
        # dense: every global qubit of the gate trades places with a local one, then the gate is local
        moved = self._bring_local([q for q in (qa, qb) if q >= k], exclude=[q for q in (qa, qb) if q < k])
        it = iter(moved)
        self._queue_local(([next(it) if qa >= k else qa, next(it) if qb >= k else qb], U))

    def _bring_local(self, glob, exclude) -> list[int]:
        """Swap-and-stay: the global actual bits `glob` trade places with local bits that are not needed
        for the longest time; returns the local bits that now hold them.  Half a shard crosses one link for
        one bit (3/4 over three links for two) and nothing is sent back: `_dyn` records the move."""
        victims = self._pick_victims(len(glob), set(exclude))
        self._trace_kind = "swap-and-stay"
        self.relayout([[v, g] for v, g in zip(victims, glob)])
        self._trace_kind = "relayout"
        # an UNPLANNED move: the contents of the swapped actual positions have traded places, the plans
        # do not know (a planned re-layout moves planned and actual positions alike: no update there)
        swap = {}
        for v, g in zip(victims, glob):
            swap[v], swap[g] = g, v
        self._dyn = [swap.get(a, a) for a in self._dyn]
        return victims

    def _pick_victims(self, count: int, exclude: set) -> list[int]:
        """Local actual bits whose next use in the running execution is farthest away (never: best).  Bits inside a
        128-byte line (0..2) come last: a slab over one of them moves 16 B per line (several times slower to pack,
        tools/relayout_probe.py) and cannot ride in a fused pass."""
        if self.k - len(exclude) < count:
            raise ValueError("not enough local qubits to bring a global gate local")
        planned_of = {a: p for p, a in enumerate(self._dyn)}
        next_use = {}
        want = {planned_of[b] for b in range(self.k) if b not in exclude}
        flat = self._flat
        for pos in range(self._flat_pos + 1, len(flat)):
            for q in flat[pos]:
                if q in want and q not in next_use:
                    next_use[q] = pos
            if len(next_use) == len(want):
                break
        cands = [b for b in range(self.k) if b not in exclude]
        cands.sort(key=lambda b: (b < 3, -next_use.get(planned_of[b], 1 << 60), -b))
        return cands[:count]

    def _scale(self, f) -> None:
        if self.k > 0:
            self._queue_local(([0], f * _I2))
        else:
            self._scale_single(f)

    def _scale_single(self, f) -> None:  # shard of one amplitude (toy sizes only)
        t = self.backend.tensor("state")
        z = complex(t[0].item(), t[1].item()) * complex(f)
        t[0], t[1] = z.real, z.imag

    # ---- synchronisation / measurement ------------------------------------------------------------
    def barrier(self) -> None:
        self._flush_local()
        self.backend.sync()
        self.dist.barrier()

    def max_over_ranks(self, value: float) -> float:
        t = self.torch.tensor([value], dtype=self.torch.float64)
        if self.dist.get_backend() == "nccl":
            t = t.cuda()
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def profile_begin(self) -> None:
        if hasattr(self.backend, "profile_begin"):
            self.backend.profile_begin()

    def profile_end(self):
        return self.backend.profile_end() if hasattr(self.backend, "profile_end") else []

    def closed_form_error(self, kind: str) -> float:
        """max over ALL 2^n amplitudes of |amp - closed form| ("ghz" / "ghz_qft"),
        evaluated on every shard in its current (staged) layout and max-reduced over ranks."""
        self._flush_local()
        local = self.backend.closed_form_error(kind, self.n, self.rank << self.k, self.l2p)
        return self.max_over_ranks(local)

    def closed_form_sample_error(self, kind: str, windows: int = 16, window: int = 256, seed: int = 7) -> float:
        """The sampled HOST check next to `closed_form_error`: `windows` runs of `window` amplitudes
        of every shard (its first and last run and random ones) are downloaded and compared on the host with the closed
        forms of SURVEY 8c written out in numpy -- an implementation that shares nothing with the device reduction;
        max over the samples of all ranks."""
        self._flush_local()
        size = 1 << self.k
        window = min(window, size)
        rng = np.random.default_rng(seed + self.rank)
        starts = sorted({0, size - window} | {int(x) for x in rng.integers(0, size - window + 1, size=max(0, windows - 2))})
        l2p = self.l2p
        worst = 0.0
        for start in starts:
            got = self.backend.download(start, window)
            x = (self.rank << self.k) + start + np.arange(window, dtype=np.int64)       # physical index
            y = np.zeros_like(x)                                                          # logical index
            for q, pbit in enumerate(l2p):
                y |= ((x >> pbit) & 1) << q
            if kind == "ghz":
                want = np.where((y == 0) | (y == (1 << self.n) - 1), 2.0 ** -0.5, 0.0).astype(np.complex128)
            elif kind == "ghz_qft":       # psi[y] = 2^-(n+1)/2 (1 + exp(-2 pi i y / 2^n))
                want = 2.0 ** (-(self.n + 1) / 2) * (1.0 + np.exp(-2j * np.pi * (y.astype(np.float64) / 2.0 ** self.n)))
            else:
                raise ValueError("kind must be 'ghz' or 'ghz_qft'")
            worst = max(worst, float(np.max(np.abs(got - want))))
        return self.max_over_ranks(worst)

    # ---- amplitude-level check of states too large to gather ------------------------------
    def shard_selectors(self) -> list:
        """(sel_mask, sel_value) per rank: the set of LOGICAL indices rank r holds in the current layout -- the logical
        qubits that sit on rank bits, with the values r gives them (qsim_fingerprint's filter on a one-GPU state)."""
        l2p = self.l2p
        glob = [q for q in range(self.n) if l2p[q] >= self.k]
        mask = sum(1 << q for q in glob)
        return [(mask, sum(((r >> (l2p[q] - self.k)) & 1) << q for q in glob)) for r in range(self.world)]

    def fingerprints(self, seed: int = 0) -> list:
        """Collective: every shard's fingerprint sum_i amp_i w(logical index of i) in its current (staged, moved) layout,
        gathered on every rank (qsim_fingerprint: evaluated on the device, two doubles per rank cross the links)."""
        self._flush_local()
        z = self.backend.fingerprint(self.n, self.rank << self.k, self.l2p, seed)
        t = self.torch.tensor([z.real, z.imag], dtype=self.torch.float64)
        if self.dist.get_backend() == "nccl":
            t = t.cuda()
        parts = [self.torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(parts, t)
        return [complex(float(p[0]), float(p[1])) for p in parts]

    def _reference_fingerprints(self, cd: dict, selector_sets: list, seed: int) -> list:
        """Rank 0 only: the circuit on ONE device (the whole 2^n state next to the shard: n = 33 is 128 GiB of the 288),
        gate semantics and order of ref_dense.simulate (ref_dense.py:44-57) through the fused single-GPU path, then the
        fingerprint of every selector's index set.  Tests with a CPU backend replace this hook."""
        from quantum_simulations_amd.runner.engine import SingleGpuEngine
        if hasattr(self.backend, "release_buffers"):
            self.backend.release_buffers()
        one = SingleGpuEngine(self.n, device=self.backend.device, mode=self.mode)
        try:
            one.init_zero_state()
            one.execute(one.plan(cd))
            # (the one-GPU engine holds its state in a layout of its own choice: undone by the fingerprint itself)
            return [[one.state.fingerprint(self.n, 0, one.l2p, seed, m, v) for m, v in sel] for sel in selector_sets]
        finally:
            one.close()

    def check_against_single_device(self, cd: dict, runs: list, seed: int = 20260504) -> list:
        """Collective.  runs = [(fingerprints, selectors), ...] taken after executions of `cd` on the partition (one per
        layout / staging variant); returns per run max_r |shard fingerprint r - the same index set of a one-device run|.
        The one-device run happens once, on rank 0; the other ranks wait in the broadcast."""
        want = [None]
        if self.rank == 0:
            want = [self._reference_fingerprints(cd, [sel for _, sel in runs], seed)]
        self.dist.broadcast_object_list(want, src=0)
        return [max(abs(g - w) for g, w in zip(got, ref)) for (got, _), ref in zip(runs, want[0])]

    # ---- BASELINE configs 4 and 5 on this engine (bench.py at N > 1, tools/run_config.py) ----------------
    def _timed_circuit(self, cd: dict):
        """(seconds of ONE execution from |0..0>, steps of the plan); `last_plan_seconds` = the host time of the plan made for
        it -- start layouts, stage boundaries and tile passes -- which a one-shot run pays in front of the execution
        (it can exceed a GHZ's execution; the records carry both)."""
        import time
        self.init_zero_state()
        self.reset_comm_stats()
        self.barrier()
        t0 = time.perf_counter()
        plan = self.plan(cd)
        self.last_plan_seconds = self.max_over_ranks(time.perf_counter() - t0)
        self.barrier()
        t0 = time.perf_counter()
        self.execute(plan)
        self.barrier()
        dt = self.max_over_ranks(time.perf_counter() - t0)
        return dt, len(plan.executions[0])

    def run_baseline_configs(self, gen, check_amplitudes: bool = True) -> dict:
        """Config 5: n-qubit GHZ and GHZ+QFT, EVERY amplitude against the closed forms of SURVEY 8c on the
        devices (max-abs-error over all shards, staged layout included).  Config 4: the seeded Clifford+T
        circuit (depth 60) with and without staging: gate-applications/s, bytes over xGMI, exchange time -- and, like
        the random 1q+CX circuit of the timed region ("checked against a 1-GPU run"), its amplitudes
        against a one-GPU run of the same circuit through layout-aware per-shard fingerprints."""
        n = self.n
        out = {"config5": [], "config4": None}
        for kind, cd in (("ghz", gen.generate_ghz_circuit(n)), ("ghz_qft", gen.generate_ghz_qft(n))):
            dt, steps = self._timed_circuit(cd)
            err = self.closed_form_error(kind)
            err_host = self.closed_form_sample_error(kind)
            out["config5"].append({"circuit": kind, "n_qubits": n, "n_gpus": self.world, "gates": len(cd["gates"]), "layout": self.layout_info,
                                   "seconds": round(dt, 4), "gate_apps_per_s": round(len(cd["gates"]) / dt, 1),
                                   "plan_seconds": round(self.last_plan_seconds, 4),
                                   "gate_apps_per_s_incl_planning": round(len(cd["gates"]) / (dt + self.last_plan_seconds), 1),
                                   "steps": steps, "max_abs_err_vs_closed_form": err, "max_abs_err_sampled_host_check": err_host,
                                   "pass_1e-10": bool(err < 1e-10 and err_host < 1e-10),
                                   "norm2": self.norm2(), "xgmi": self.comm_stats()})
        seed = 20260504
        saved = self.staging
        for key, cd, variants in (("config4", gen.random_clifford_t_circuit(n, depth=60), (("staged", True), ("unstaged", False))),
                                  ("random_1q_cx", gen.random_1q_cx_circuit(n, depth=40), (("staged", True),))):
            rec = {"n_qubits": n, "n_gpus": self.world, "gates": len(cd["gates"]), "local_qubits": self.k}
            runs = []
            for label, staging in variants:
                self.staging = staging
                dt, steps = self._timed_circuit(cd)
                stats = self.comm_stats()
                ms = stats.get("exchange_ms_max_over_ranks")
                rec[label] = {"seconds": round(dt, 4), "gate_apps_per_s": round(len(cd["gates"]) / dt, 1), "steps": steps,
                              "plan_seconds": round(self.last_plan_seconds, 4),
                              "gate_apps_per_s_incl_planning": round(len(cd["gates"]) / (dt + self.last_plan_seconds), 1),
                              "hbm_passes": self.last_passes, "norm2": self.norm2(), "xgmi": stats,
                              # device-side exchange time (stream events, max over ranks; RCCL runs
                              # only) over the run's wall time -- pieces overlap compute, so this is an upper bound of
                              # what the links cost
                              "exchange_time_share": (round(ms * 1e-3 / dt, 4) if ms is not None else None),
                              "layout": self.layout_info}
                runs.append((self.fingerprints(seed), self.shard_selectors()))
            self.staging = saved
            if check_amplitudes:
                # EVERY amplitude of the partitioned result enters its shard's fingerprint; the same index sets of a one-GPU
                # run of the same circuit must give the same sums (north star: "matching reference amplitudes to 1e-10")
                diffs = self.check_against_single_device(cd, runs, seed)
                for (label, _), d in zip(variants, diffs):
                    rec[label]["fingerprint_max_abs_diff_vs_single_gpu"] = d
                    rec[label]["pass_1e-10"] = bool(d < 1e-10)
            out[key] = rec
        return out

    def measure_relayouts(self, reps: int = 2) -> list:
        """Collective.  What an all-to-all over m rank bits costs on THIS machine, m = 1 .. p, measured on the idle shard
        (whatever state it holds: every re-layout is made twice, there and back, so the state and the layout are the same
        afterwards): local bits k-1, k-2, ... trade places with rank bits 0, 1, ...  With nothing queued around it a
        re-layout is a pack pass, the exchange (piece by piece, all peers at once) and an unpack pass; reported per m:
        host-clock milliseconds (max over ranks) of the whole thing, device-event milliseconds of the exchange alone (RCCL
        runs: stream events from the first post to the last arrival; max over ranks), bytes per rank and the rate they
        give.  Replaces the modelled RELAYOUT_PASSES once a multi-GPU node has run it (bench.py prints
        `relayout_in_pass_units`)."""
        import time
        out = []
        shard_bytes = 16 << self.k
        for m in range(1, self.p + 1):
            pairs = [[self.k - 1 - i, self.k + i] for i in range(m)]
            if self.k - m < 3:
                break
            wall, events = [], []
            for _ in range(2 * max(1, reps)):
                self._flush_local()
                self.barrier()
                self.reset_comm_stats()
                t0 = time.perf_counter()
                self.relayout(pairs)
                self._flush_local()
                self.barrier()
                wall.append(self.max_over_ranks(time.perf_counter() - t0) * 1e3)
                events.append(self.comm_stats().get("exchange_ms_max_over_ranks"))
            sent = shard_bytes - (shard_bytes >> m)
            ev = [e for e in events if e]
            best_ev = min(ev) if ev else None
            out.append({"m": m, "local_bits": [pr[0] for pr in pairs], "bytes_sent_per_rank": sent, "pieces": self._relayout_pieces(self.k - m),
                        "wall_ms_pack_exchange_unpack": round(min(wall), 3), "exchange_event_ms": None if best_ev is None else round(best_ev, 3),
                        "exchange_GBps_per_rank": None if best_ev is None else round(sent / (best_ev * 1e-3) / 1e9, 1),
                        "wall_GBps_per_rank": round(sent / (min(wall) * 1e-3) / 1e9, 1), "repeats": 2 * max(1, reps)})
        self.reset_comm_stats()
        return out

    def comm_stats(self) -> dict:
        """Collective (every rank calls it): bytes and exchanges of this rank, device-side exchange time of this rank
        and the maximum over all ranks (RCCL runs only: stream events around every exchange)."""
        ms = ms_max = None
        if self._comm_events:
            self.backend.sync()
            ms = float(sum(a.elapsed_time(b) for a, b in self._comm_events))
        if not self.dry and self.dist.is_initialized() and self.dist.get_backend() == "nccl":
            ms_max = self.max_over_ranks(ms or 0.0)
        return {"bytes_sent_per_rank": self.xgmi_bytes_sent, "exchanges": self.exchanges,
                "exchange_ms_rank0": ms, "exchange_ms_max_over_ranks": ms_max}

    def reset_comm_stats(self) -> None:
        self.xgmi_bytes_sent, self.exchanges, self._comm_events, self.relayout_log = 0, 0, [], []

    def close(self) -> None:
        self.backend.close()
        if self.dist.is_initialized():
            self.dist.barrier()
            self.dist.destroy_process_group()
