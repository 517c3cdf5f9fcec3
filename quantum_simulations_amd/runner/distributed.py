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
  other gates on a global qubit ..... full-shard exchange with the partner rank over one xGMI
                                       link + the partner-chunk kernel (apply_*_pair semantics)
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

    def download(self) -> np.ndarray:
        return self.chunk("state").download()

    def sync(self) -> None:
        self.torch.cuda.synchronize(self.device)

    # ---- arithmetic -----------------------------------------------------------------
    def apply_ops(self, ops) -> int:
        return self.chunk("state").apply_ops(ops)      # HBM passes (fused tile launches)

    def apply_1q_pair(self, names, U) -> None:
        from quantum_simulations_amd.kernel import gpu_nonlocal
        gpu_nonlocal.apply_1q_pair(self.chunk(names[0]), self.chunk(names[1]), U)

    def apply_2q_pair_qa_local(self, names, qa, U) -> None:
        from quantum_simulations_amd.kernel import gpu_nonlocal
        gpu_nonlocal.apply_2q_pair_qa_local(self.chunk(names[0]), self.chunk(names[1]), qa, U)

    def apply_2q_pair_qb_local(self, names, qb, U) -> None:
        from quantum_simulations_amd.kernel import gpu_nonlocal
        gpu_nonlocal.apply_2q_pair_qb_local(self.chunk(names[0]), self.chunk(names[1]), qb, U)

    def apply_2q_quad(self, names, U) -> None:
        from quantum_simulations_amd.kernel import gpu_nonlocal
        gpu_nonlocal.apply_2q_quad(*(self.chunk(n) for n in names), U)

    def pack_bits(self, bits, pattern: int, dst: str, dst_offset: int) -> None:
        self.chunk("state").pack_bits(bits, pattern, self.chunk(dst), dst_offset)

    def unpack_bits(self, bits, pattern: int, src: str, src_offset: int) -> None:
        self.chunk("state").unpack_bits(bits, pattern, self.chunk(src), src_offset)

    def pack_all(self, bits, dst: str, skip_pattern: int, piece: int = 0, n_pieces: int = 1) -> None:
        self.chunk("state").pack_all(bits, self.chunk(dst), skip_pattern, piece, n_pieces)

    def unpack_all(self, bits, src: str, skip_pattern: int, piece: int = 0, n_pieces: int = 1) -> None:
        self.chunk("state").unpack_all(bits, self.chunk(src), skip_pattern, piece, n_pieces)

    def closed_form_error(self, kind: str, n_total: int, base_index: int, log_to_phys) -> float:
        return self.chunk("state").max_abs_err_closed_form(kind, n_total, base_index, log_to_phys)

    def profile_begin(self) -> None:
        self.chunk("state").profile_begin()

    def profile_end(self):
        return self.chunk("state").profile_end()

    def close(self) -> None:
        self.sync()
        for c in self._chunks.values():
            c.close()
        self._chunks.clear()
        self._tensors.clear()


class Plan:
    """Step lists for successive executions (the staging layout carries over between them)."""

    def __init__(self, executions: list, mappings: list):
        self.executions, self.mappings, self.cursor = executions, mappings, 0


class DistributedEngine:
    def __init__(self, n_qubits: int, world: int, rank: int, local_rank: int = 0,
                 mode: str = "fused", backend=None, staging: bool = True,
                 staging_method: str = "belady", init_process_group: bool = True,
                 relayout_pieces: int = 4, min_piece_qubits: int = 20):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        if world & (world - 1) or world < 2:
            raise ValueError("world size must be a power of two >= 2")
        self.n, self.world, self.rank = n_qubits, world, rank
        self.p = world.bit_length() - 1
        self.k = n_qubits - self.p
        if self.k < 0:
            raise ValueError("more ranks than amplitudes")
        self.mode, self.staging, self.staging_method = mode, staging, staging_method
        import os
        rehearsal = backend is None and os.environ.get("QSIM_DIST_BACKEND") == "gloo"
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
        self.backend = backend if backend is not None else HipShardBackend(self.k, local_rank)
        self.l2p = list(range(n_qubits))      # logical qubit -> physical index bit
        self.xgmi_bytes_sent = 0
        self.exchanges = 0
        self._comm_events: list = []
        if relayout_pieces not in (1, 2, 4, 8):
            raise ValueError("relayout_pieces must be 1, 2, 4 or 8")
        self.relayout_pieces, self.min_piece_qubits = relayout_pieces, min_piece_qubits
        self._passes = self.last_passes = 0
        self._pending: list = []

    # ---- helpers -----------------------------------------------------------------------
    def _rank_bit(self, phys_qubit: int) -> int:
        return (self.rank >> (phys_qubit - self.k)) & 1

    def _partner(self, phys_qubit: int) -> int:
        return self.rank ^ (1 << (phys_qubit - self.k))

    def _post(self, transfers):
        """Post [(peer, send_tensor, recv_tensor)] together, without waiting (RCCL: the transfer is
        ordered after everything already queued on the current stream)."""
        dist, torch = self.dist, self.torch
        staged, ops = [], []
        host = dist.get_backend() == "gloo"
        for peer, send, recv in transfers:
            if host and send.is_cuda:  # rehearsal of several ranks on one GPU: stage through host
                s_h, r_h = send.cpu(), torch.empty(recv.shape, dtype=recv.dtype)
                staged.append((recv, r_h))
                send, recv = s_h, r_h
            ops.append(dist.P2POp(dist.isend, send, peer))
            ops.append(dist.P2POp(dist.irecv, recv, peer))
            self.xgmi_bytes_sent += send.numel() * send.element_size()
        return (dist.batch_isend_irecv(ops) if ops else [], staged)

    @staticmethod
    def _finish(posted) -> None:
        """Received data may be used by what is queued after this (RCCL: the current stream waits,
        not the host)."""
        works, staged = posted
        for work in works:
            work.wait()
        for dev_t, host_t in staged:
            dev_t.copy_(host_t)

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

    def _exchange(self, transfers) -> None:
        """transfers: [(peer, send_tensor, recv_tensor)], all posted together, then awaited."""
        if not transfers:
            return
        timer = self._comm_timer(transfers[0][1])
        self._finish(self._post(transfers))
        self._comm_done(timer)

    # ---- deferred local work -------------------------------------------------------------------
    # Global-qubit gates that need no exchange end up as LOCAL ops on this rank (a rank-bit phase,
    # a diagonal on the local partner qubit, the conditional 1q gate of a global control).  Run one
    # by one each is a full HBM pass of the shard; queued, they ride in the next fused local pass
    # (program order is kept: they sit between two local batches), or are flushed before anything
    # that reads the shard (an exchange, a re-layout, a reduction, a download).
    def _queue_local(self, op) -> None:
        self._pending.append(op)

    def _flush_local(self) -> None:
        if self._pending:
            ops, self._pending = self._pending, []
            self._passes += self.backend.apply_ops(ops) or 0

    # ---- state ---------------------------------------------------------------------------
    def init_zero_state(self) -> None:
        self._pending = []
        self.backend.init_zero(self.rank == 0)
        self.l2p = list(range(self.n))

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
    def _steps_from(self, cd: dict, l2p: list[int]):
        """Plan `cd` for a state whose logical qubit q currently sits at physical bit l2p[q]."""
        relabeled = {"number_of_qubits": self.n,
                     "gates": [{"qubits": [l2p[q] for q in g["qubits"]], "gate": g["gate"],
                                "params": g["params"]} for g in cd["gates"]]}
        if self.staging and self.k >= 2:   # (staging cannot hold a 2-qubit gate in fewer than 2 local qubits)
            steps, moved = atlas_stages(relabeled, self.k, method=self.staging_method,
                                        strict_order=True)
        else:
            steps, moved = batch_levels(levelize(relabeled), self.k), list(range(self.n))
        return steps, [moved[l2p[q]] for q in range(self.n)]

    def plan(self, circuit_dict: dict, repeats: int = 1) -> Plan:
        cd = validate_circuit_dict(circuit_dict)
        if cd["number_of_qubits"] != self.n:
            raise ValueError(f"circuit has {cd['number_of_qubits']} qubits, engine has {self.n}")
        executions, mappings = [], []
        l2p = list(self.l2p)
        for _ in range(max(1, repeats)):
            steps, l2p = self._steps_from(cd, l2p)
            executions.append(steps)
            mappings.append(list(l2p))
        return Plan(executions, mappings)

    def passes_per_step(self, plan: Plan) -> int:
        """HBM passes of the last executed circuit on this rank: fused tile launches of the local
        steps + 2 per re-layout (pack, unpack); before any execution, the op count of the plan."""
        if self.last_passes:
            return self.last_passes
        return sum(len(s["local_ops"]) + len(s["nonlocal_ops"]) for s in plan.executions[0])

    # ---- execution ---------------------------------------------------------------------------
    def execute(self, plan: Plan) -> None:
        i = plan.cursor
        if i >= len(plan.executions):
            raise RuntimeError("plan exhausted: call engine.plan(circuit, repeats=K) with enough repeats")
        self._passes = 0
        for step in plan.executions[i]:
            self.run_step(step)
        self._flush_local()
        self.last_passes = self._passes
        self.l2p = list(plan.mappings[i])
        plan.cursor = i + 1

    def run_step(self, step: dict) -> None:
        if step["local_ops"]:
            ops, self._pending = self._pending + list(step["local_ops"]), []
            self._passes += self.backend.apply_ops(ops) or 0
        ops = step["nonlocal_ops"]
        i = 0
        while i < len(ops):
            j = i
            group = []
            used: set[int] = set()
            while j < len(ops) and self._is_relayout_swap(ops[j]) and used.isdisjoint(ops[j][0]):
                group.append(ops[j][0])
                used.update(ops[j][0])
                j += 1
            if group:
                self.relayout(group)
                self._passes += 2
                i = j
            else:
                self.apply_nonlocal(*ops[i])
                i += 1

    def _is_relayout_swap(self, op) -> bool:
        qs, U = op
        return (len(qs) == 2 and (qs[0] < self.k) != (qs[1] < self.k) and U.shape == (4, 4)
                and np.array_equal(U, _SWAP))

    # -- all-to-all re-layout: swap m local bits with m global bits ------------------------------
    def relayout(self, pairs) -> None:
        """pairs: [[p_a, p_b], ...] each with exactly one local and one global physical bit.

        Pipelined in `pieces` sub-ranges of every slab: while piece s is on the links, piece s+1 is
        being packed and piece s-1 unpacked (the pack / unpack passes are 10-30 % of a re-layout's
        time at 8 GPUs when run back to back)."""
        loc = [min(p) for p in pairs]
        glo = [max(p) for p in pairs]
        m = len(pairs)
        slab = 2 << (self.k - m)                       # float64 elements per slab
        mine = sum(((self.rank >> (g - self.k)) & 1) << i for i, g in enumerate(glo))
        send, recv = self.backend.tensor("buf0"), self.backend.tensor("buf1")
        peers = []
        for d in range(1 << m):
            if d == mine:
                continue
            peer = self.rank
            for i, g in enumerate(glo):
                peer = (peer & ~(1 << (g - self.k))) | (((d >> i) & 1) << (g - self.k))
            peers.append((d, peer))
        self._flush_local()
        pieces = self._relayout_pieces(self.k - m)
        part = slab // pieces
        timer = self._comm_timer(send)
        posted = []
        self.backend.pack_all(loc, "buf0", mine, 0, pieces)        # slab d at offset d * 2^(k-m)
        for s in range(pieces):
            posted.append(self._post([(peer, send[d * slab + s * part:d * slab + (s + 1) * part],
                                       recv[d * slab + s * part:d * slab + (s + 1) * part]) for d, peer in peers]))
            if s + 1 < pieces:
                self.backend.pack_all(loc, "buf0", mine, s + 1, pieces)
        for s in range(pieces):
            self._finish(posted[s])
            self.backend.unpack_all(loc, "buf1", mine, s, pieces)
        self._comm_done(timer)

    def _relayout_pieces(self, slab_qubits: int) -> int:
        """Pieces per slab: the configured count, capped so that a piece keeps >= 2^20 amplitudes
        (16 MiB per peer and piece -- large enough for full link rate)."""
        want = self.relayout_pieces
        while want > 1 and slab_qubits - (want.bit_length() - 1) < self.min_piece_qubits:
            want //= 2
        return max(1, want)

    # -- one gate with at least one global qubit ---------------------------------------------------
    def apply_nonlocal(self, qs, U) -> None:
        k = self.k
        if len(qs) == 1:
            q = qs[0]
            b = self._rank_bit(q)
            if _is_diagonal(U):                       # rank-bit phase, no exchange
                if U[b, b] != 1:
                    self._scale(U[b, b])
                return
            self._exchange_full(self._partner(q))
            names = ("state", "buf1") if b == 0 else ("buf1", "state")
            self.backend.apply_1q_pair(names, U)
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
            if self._rank_bit(ctrl):
                if tgt < k:
                    self._queue_local(([tgt], V))
                else:
                    self.apply_nonlocal([tgt], V)
            return
        if a_glob and b_glob:                         # dense, both global: group of four ranks
            self._quad(qa, qb, U)
            return
        if a_glob:                                    # qa = partner bit (MSB), qb local
            self._exchange_full(self._partner(qa))
            names = ("state", "buf1") if self._rank_bit(qa) == 0 else ("buf1", "state")
            self.backend.apply_2q_pair_qb_local(names, qb, U)
        else:                                         # qa local, qb = partner bit (LSB)
            self._exchange_full(self._partner(qb))
            names = ("state", "buf1") if self._rank_bit(qb) == 0 else ("buf1", "state")
            self.backend.apply_2q_pair_qa_local(names, qa, U)

    def _scale(self, f) -> None:
        if self.k > 0:
            self._queue_local(([0], f * _I2))
        else:
            self._scale_single(f)

    def _scale_single(self, f) -> None:  # shard of one amplitude (toy sizes only)
        t = self.backend.tensor("state")
        z = complex(t[0].item(), t[1].item()) * complex(f)
        t[0], t[1] = z.real, z.imag

    def _exchange_full(self, peer: int) -> None:
        """Partner's whole shard into buf1 (apply_*_pair needs both chunks)."""
        self._flush_local()
        self._exchange([(peer, self.backend.tensor("state"), self.backend.tensor("buf1"))])

    def _quad(self, qa: int, qb: int, U) -> None:
        self._flush_local()
        ba, bb = self._rank_bit(qa), self._rank_bit(qb)
        me = 2 * ba + bb
        names = [None] * 4
        transfers = []
        free = ["buf1", "buf2", "buf3"]
        for idx in range(4):
            if idx == me:
                names[idx] = "state"
                continue
            peer = self.rank
            peer = (peer & ~(1 << (qa - self.k))) | ((idx >> 1) << (qa - self.k))
            peer = (peer & ~(1 << (qb - self.k))) | ((idx & 1) << (qb - self.k))
            names[idx] = free.pop(0)
            transfers.append((peer, self.backend.tensor("state"), self.backend.tensor(names[idx])))
        self._exchange(transfers)
        self.backend.apply_2q_quad(names, U)

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

    def comm_stats(self) -> dict:
        ms = None
        if self._comm_events:
            self.backend.sync()
            ms = float(sum(a.elapsed_time(b) for a, b in self._comm_events))
        return {"bytes_sent_per_rank": self.xgmi_bytes_sent, "exchanges": self.exchanges,
                "exchange_ms_rank0": ms}

    def reset_comm_stats(self) -> None:
        self.xgmi_bytes_sent, self.exchanges, self._comm_events = 0, 0, []

    def close(self) -> None:
        self.backend.close()
        if self.dist.is_initialized():
            self.dist.barrier()
            self.dist.destroy_process_group()
