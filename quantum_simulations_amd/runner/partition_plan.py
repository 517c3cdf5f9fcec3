"""Stage boundaries and fused tile passes of a partitioned state planned TOGETHER (staging method "tiles").

The reference plans stages first and hands each stage's gates to the kernels afterwards (wenbo_engine/circuit/
staging.py:447-519 `_local_sets_to_steps`: every gate runs in the FIRST stage whose local set holds it; fusion.py:86-142
batches what a stage got).  On a GPU shard a stage is executed as fused tile passes -- one HBM round trip of the shard for
the ~45 ops whose targets meet in a tile of 11 index bits -- and stages cut that way end in passes that hold a handful of
ops: the cone of gates a stage can reach narrows towards its end.  33 qubits on 8 ranks needed 33 passes for the circuit
that takes 19 on one 30-qubit GPU.

Here the pass builder itself walks the whole circuit.  `qsim_plan_peek_pass` (csrc/tile_planner.h, peek mode) answers "which
tile would the next pass take, and which ops would it hold" for a partly executed op list on a partitioned state: index bits
>= k are rank bits -- fine as controls and phase bits (a rank applies or skips such an op by its own bits: no exchange),
never targets.  The planner commits passes while they are worth a round trip and calls for a re-layout when the best next
pass is thin (fewer than `min_ops` ops): the ops it would have held stay in the pool and ride in the passes AFTER the
re-layout together with what the incoming qubits unblock -- a gate that is executable on either side of a re-layout goes
where a pass has room.  The global set is chosen by farthest next use as a TARGET (Belady; a control or phase use of a
global qubit costs nothing), never among the tile bits of the pass that stores the slabs (a slab bit inside that tile would
cost a separate pack pass), never among the qubits on the three line bits (a slab bit inside a 128-byte line cannot ride in
a tile pass), and a re-layout swaps ALL rank bits when it can (2^-m of a shard per link: wide is cheap).  `min_ops` is not
guessed: `plan_partition_best` runs the planner for several values in parallel threads (the library call releases the GIL)
and keeps the schedule with the lowest cost = passes + re-layouts in pass units.

The result has the reference's step format (`local_ops` / `nonlocal_ops` with SWAP lists as re-layout requests), so
`DistributedEngine.run_step` executes it unchanged; the first step of every segment carries `tile_masks`, the tiles the
planner chose, which the engine hands to the library (qsim_ops_io::tile_masks) so that every rank runs the passes that
were planned whatever its own rank-bit conditions removed from the list.
Host only: needs libqsim_hip.so (the pass builder is host code of the library), no device.
"""
from __future__ import annotations

import ctypes as C
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from quantum_simulations_amd.kernel import gates as gate_table

_SWAP = gate_table.SWAP()
_I2 = np.eye(2, dtype=np.complex128)
LINE_BITS = 3          # index bits 0..2: one 128-byte line, members of every tile
MIN_OPS_CHOICES = (8, 16, 20, 24, 28, 32)
# An all-to-all over m rank bits in units of one fused pass of the shard (DistributedEngine.RELAYOUT_PASSES: a model until a
# multi-GPU node measures it; bench.py --gpus N prints the measured ratios)
RELAYOUT_COST = {1: 10.4, 2: 5.2, 3: 2.6}


def op_targets(qubits, U) -> list:
    """The qubits an op acts on NON-diagonally (they must be local index bits when it runs); controls and phase bits may
    be rank bits (DistributedEngine.apply_nonlocal: rank-bit phase / conditional gate, no exchange)."""
    if len(qubits) == 1:
        return [] if not (U[0, 1] or U[1, 0]) else [qubits[0]]
    if not np.any(U - np.diag(np.diag(U))):
        return []
    if np.array_equal(U[:2, :2], _I2) and not np.any(U[:2, 2:]) and not np.any(U[2:, :2]):
        return [qubits[1]]                                       # control = qubits[0]
    P = U[np.ix_([0, 2, 1, 3], [0, 2, 1, 3])]
    if np.array_equal(P[:2, :2], _I2) and not np.any(P[:2, 2:]) and not np.any(P[2:, :2]):
        return [qubits[0]]                                       # control = qubits[1]
    return list(qubits)


class PackedOps:
    """An op list packed once for many planning runs: arities, LABELS (the index bits the qubits start on), matrices,
    targets per op."""

    def __init__(self, ops, n: int):
        from quantum_simulations_amd.kernel.device import pack_ops
        self.ops, self.n = ops, n
        self.nq, self.labels, self.mats = pack_ops(ops)
        self.targets = [op_targets(qs, U) for qs, U in ops]
        self.one_q = self.nq == 1
        # exact identities (H H, S^4 ... after 1q fusion): the library's classifier drops them, no pass ever holds them
        self.identity = np.array([bool(np.array_equal(U, np.eye(U.shape[0]))) for _, U in ops], dtype=bool)

    def relabeled(self, l2p) -> "PackedOps":
        """The same op list with qubit q moved to index bit l2p[q] (matrices, arities and op order shared)."""
        other = object.__new__(PackedOps)
        other.n, other.nq, other.mats, other.one_q, other.identity = self.n, self.nq, self.mats, self.one_q, self.identity
        m = np.asarray(l2p, dtype=np.int32)
        other.labels = np.ascontiguousarray(m[self.labels])
        other.labels[1::2][self.one_q] = 0
        other.ops = [([int(m[q]) for q in qs], U) for qs, U in self.ops]
        other.targets = [[int(m[q]) for q in t] for t in self.targets]
        return other


def _peek(lib, check, packed: PackedOps, k: int, cur, done, members, avoid: int = 0, cache: dict | None = None):
    """(tile mask, needed bits, member ops) of the next pass.  `cache`: planner runs that differ only in their thin-pass
    threshold make the same decisions up to the first pass whose size lies between the thresholds: the same (layout, done
    set) is asked again and again."""
    key = None
    if cache is not None:
        key = (cur.tobytes(), done.tobytes(), avoid)
        hit = cache.get(key)
        if hit is not None:
            return hit
    out = _peek_uncached(lib, check, packed, k, cur, done, members, avoid)
    if cache is not None:
        cache[key] = out
    return out


def _peek_uncached(lib, check, packed: PackedOps, k: int, cur, done, members, avoid: int = 0):
    qs = np.ascontiguousarray(cur[packed.labels])
    qs[1::2][packed.one_q] = 0
    mask, need, count = C.c_uint64(), C.c_uint64(), C.c_int32()
    check(lib.qsim_plan_peek_pass(k, packed.n, len(packed.nq), packed.nq.ctypes.data_as(C.c_void_p), qs.ctypes.data_as(C.c_void_p),
                                  packed.mats.ctypes.data_as(C.c_void_p), done.ctypes.data_as(C.c_void_p), avoid, 0,
                                  C.byref(mask), C.byref(need), C.byref(count), members.ctypes.data_as(C.c_void_p)))
    return int(mask.value), int(need.value), [int(i) for i in members[:count.value]]


def plan_partition(ops, n: int, k: int, min_ops: int = 20, full_width: bool = True, relayout_cost=None, cache: dict | None = None) -> dict:
    """ops: [(qubits, U)] (or a PackedOps) on index bits 0..n-1 of the whole state (bits >= k are rank bits), program order.
    -> {"steps", "moved", "passes", "relayouts", "cost", "segments", "min_ops"}: steps in the reference's format in the
    index bits of their time, moved[b] = where the qubit that started on bit b ends, passes = fused passes committed,
    relayouts = [m, ...], cost = passes + re-layouts in pass units, segments = [{"ops", "passes", "tile_masks"}, ...]."""
    from quantum_simulations_amd import _lib
    lib, check = _lib.load(), _lib.check
    packed = ops if isinstance(ops, PackedOps) else PackedOps(ops, n)
    ops = packed.ops
    relayout_cost = relayout_cost or RELAYOUT_COST
    p = n - k
    n_ops = len(ops)
    done = packed.identity.astype(np.uint8)                      # (identities are dropped from the schedule)
    cur = np.arange(n, dtype=np.int32)                           # label -> index bit it sits on now
    scratch = np.zeros(max(1, n_ops), dtype=np.int32)
    segments = [{"idx": [], "masks": [], "needs": [], "relayout": None}]
    never = n_ops + 1

    def next_target_use() -> list:
        use = [never] * n
        left = n
        for i in range(n_ops):
            if not done[i]:
                for q in packed.targets[i]:
                    if use[q] == never:
                        use[q] = i
                        left -= 1
                if not left:
                    break
        return use

    def commit(mask, need, members) -> None:
        done[members] = 1
        seg = segments[-1]
        seg["idx"] += members
        seg["masks"].append(mask)
        seg["needs"].append(need)

    while not done.all():
        mask, need, members = _peek(lib, check, packed, k, cur, done, scratch, cache=cache)
        use = next_target_use()
        waiting = any(use[q] != never for q in range(n) if cur[q] >= k)      # an op somewhere waits for a rank bit
        fresh_segment = len(segments) > 1 and not segments[-1]["masks"]      # (a re-layout right behind a re-layout buys nothing)
        last_need = segments[-1]["needs"][-1] if segments[-1]["needs"] else 0
        # The LAST re-layout can come as early as p local qubits are finished (never a target again): they leave, every
        # global qubit comes in, and nothing will ever wait for a rank bit again -- the ops that wait now join the passes
        # from here on instead of filling thin ones at the end (GHZ+QFT: 8 -> 7 passes at 33 qubits).
        finished = [q for q in range(n) if LINE_BITS <= cur[q] < k and use[q] == never and not (last_need >> int(cur[q])) & 1]
        final_now = waiting and not fresh_segment and len(finished) >= p and segments[-1]["masks"]
        if members and not final_now and (len(members) >= min_ops or not waiting or fresh_segment):
            commit(mask, need, members)
            continue
        # A thin pass: is there a better global set?  Farthest next use as a target goes out.  The slab bits of a fused
        # re-layout must not be tile bits of the pass that stores the slabs (the segment's last): qubits that pass needs as
        # tile bits are no candidates (its fill bits are re-chosen below); qubits on the line bits never are.
        glob = [q for q in range(n) if cur[q] >= k]
        local = [q for q in range(n) if LINE_BITS <= cur[q] < k]
        if sum(1 for q in local if not (last_need >> int(cur[q])) & 1) >= p:
            local = [q for q in local if not (last_need >> int(cur[q])) & 1]
        # (tiny shards, whose tile is the whole shard: the slab-storing pass runs in place and a pack pass follows)
        local.sort(key=lambda q: (-use[q], q))
        # the local qubits needed last go out, the global ones needed first come in -- as long as the one coming in is needed
        # before the one going out
        out, inc = [], []
        for a, b in zip(local[:p], sorted(glob, key=lambda q: (use[q], q))):
            if use[b] >= use[a]:
                break
            out.append(a)
            inc.append(b)
        if full_width and out and len(out) < p and len(local) >= p:
            # ... unless that makes the all-to-all narrower: a wide one is cheaper than a narrow one (2^-m of a shard per
            # link, the links work in parallel), so idle rank bits are swapped along
            out, inc = local[:p], list(glob)
        out.sort(key=lambda q: cur[q])
        inc.sort(key=lambda q: cur[q])
        if not out:
            if not members:
                raise RuntimeError("partition planner made no progress")      # (unreachable: a waiting target ranks first)
            commit(mask, need, members)
            continue
        pairs = [[int(cur[a]), int(cur[b])] for a, b in zip(out, inc)]        # [local bit, rank bit]
        seg = segments[-1]
        seg["relayout"] = pairs
        if seg["masks"]:                   # the last tile's fill: not the slab bits
            slab = sum(1 << a for a, _ in pairs)
            want = bin(seg["masks"][-1]).count("1")
            tile = last_need
            for b in range(LINE_BITS, k):
                if bin(tile).count("1") >= want:
                    break
                if not (slab >> b) & 1 and not (tile >> b) & 1:
                    tile |= 1 << b
            seg["masks"][-1] = tile
        for a, b in zip(out, inc):
            cur[a], cur[b] = cur[b], cur[a]
        segments.append({"idx": [], "masks": [], "needs": [], "relayout": None})
    # ---- steps in the reference's format -------------------------------------------------------------------------
    steps, seg_info, relayouts = [], [], []
    where = np.arange(n, dtype=np.int32)
    for seg in segments:
        local_run, glob_run = [], []
        first_step = len(steps)

        def close():
            if local_run or glob_run:
                steps.append({"local_ops": list(local_run), "nonlocal_ops": list(glob_run)})
            local_run.clear()
            glob_run.clear()
        for i in sorted(seg["idx"]):
            qs, U = ops[i]
            now = [int(where[q]) for q in qs]
            if max(now) < k:
                if glob_run:
                    close()
                local_run.append((now, U))
            else:
                glob_run.append((now, U))
        close()
        if len(steps) > first_step:
            steps[first_step]["tile_masks"] = list(seg["masks"])   # the tiles of the segment's passes, in order
            steps[first_step]["tile_needs"] = list(seg["needs"])   # ... and the bits of each that its ops need (the rest is fill)
        seg_info.append({"ops": len(seg["idx"]), "passes": len(seg["masks"]), "tile_masks": list(seg["masks"])})
        if seg["relayout"]:
            steps.append({"local_ops": [], "nonlocal_ops": [(pr, _SWAP) for pr in seg["relayout"]]})
            relayouts.append(len(seg["relayout"]))
            at = {int(where[q]): q for q in range(n)}
            for a, b in seg["relayout"]:
                qa, qb = at[a], at[b]
                where[qa], where[qb] = b, a
    passes = sum(s["passes"] for s in seg_info)
    return {"steps": steps, "moved": [int(x) for x in where], "passes": passes, "relayouts": relayouts,
            "cost": passes + sum(relayout_cost[m] for m in relayouts), "segments": seg_info, "min_ops": min_ops,
            "full_width": full_width}


_POOL = None


def planning_pool(threads: int = 8) -> ThreadPoolExecutor:
    """One long-lived pool for the planner's parallel runs (threads that live on get spread over the cores; a pool per call
    measured 4x slower on short jobs)."""
    global _POOL
    if _POOL is None:
        import os
        try:
            cores = len(os.sched_getaffinity(0))
        except AttributeError:
            cores = os.cpu_count() or 1
        _POOL = ThreadPoolExecutor(max_workers=max(1, min(threads, cores)), thread_name_prefix="qsim-plan")
    return _POOL


def plan_partition_best(ops, n: int, k: int, choices=MIN_OPS_CHOICES, threads: int = 8, relayout_cost=None) -> dict:
    """The cheapest schedule over the thresholds `choices` (and both re-layout widths), planned in parallel threads; ties:
    the earlier choice.  Deterministic: every rank of a run computes the same schedule from the same circuit."""
    packed = ops if isinstance(ops, PackedOps) else PackedOps(ops, n)
    jobs = [(m, True) for m in choices]          # (narrow re-layouts never won on the seeded workloads: not tried)
    # A shared cache of pass-builder answers: the runs agree up to the first pass whose size lies between two thresholds.
    # In parallel threads (the default) a run only profits from what another has already asked; one after the other
    # (threads <= 1) the later runs ask little that is new (measured on 8 cores: 25 executions at 33 qubits / 8 ranks planned
    # in 5.4 s in parallel against 11.5 s in sequence; at 31 qubits / 2 ranks, where the runs hardly differ, 15 s against 8 s).
    cache: dict = {}

    def run(job):
        return plan_partition(packed, n, k, min_ops=job[0], full_width=job[1], relayout_cost=relayout_cost, cache=cache)
    if threads > 1 and len(jobs) > 1:
        results = list(planning_pool(threads).map(run, jobs))
    else:
        results = [run(j) for j in jobs]
    best = min(range(len(results)), key=lambda i: (results[i]["cost"], i))
    out = results[best]
    out["tried"] = [{"min_ops": j[0], "full_width": j[1], "passes": r["passes"], "relayouts": r["relayouts"], "cost": round(r["cost"], 2)}
                    for j, r in zip(jobs, results)]
    return out


def place_slots(executions: list, k: int, seeds=(1, 2), fit_executions: int = 6) -> tuple:
    """Which index bit every LOCAL SLOT of a planned partition schedule should be, for the DRAM pattern of its tiles.

    A schedule of `plan_partition` is invariant under a permutation of the local index bits above the line bits, applied to
    everything at once (ops, tiles, the local side of every re-layout, the start layout): a qubit that comes in later takes
    the slot of the one that left.  What a pass costs depends on WHICH index bits its tile holds (runner/tile_layout.py: a
    model fitted to measured passes), so the permutation is chosen by annealing the model's total over the tiles' NEEDED
    bits of all executions (qsim_choose_layout), and then every tile's FILL bits -- free per pass -- are picked greedily
    under the same model (never the slab bits of the re-layout the pass stores into).  Rewrites the steps of `executions`
    (lists of steps, chained: execution i + 1 starts in the layout execution i leaves) in place.
    -> (sigma: old local bit -> new local bit over all n bits touched, model ms before, model ms after)."""
    from quantum_simulations_amd.runner import tile_layout
    model = tile_layout.model_for(k)
    needs = [m for steps in executions for st in steps for m in st.get("tile_needs", [])]
    if not needs:
        return None, 0.0, 0.0
    bits_of = lambda m: [b for b in range(LINE_BITS, k) if (int(m) >> b) & 1]          # noqa: E731
    before = sum(tile_layout.tile_cost(model, bits_of(m)) for steps in executions for st in steps for m in st.get("tile_masks", []))
    # (the annealing takes time in proportion to the tiles: a long chain of executions is fitted on its first few)
    fit = [m for steps in executions[:max(1, fit_executions)] for st in steps for m in st.get("tile_needs", [])]
    sigma_k, _c0, _c1 = min((tile_layout.choose_layout([bits_of(m) for m in fit], k, seed=sd) for sd in seeds), key=lambda r: r[2])
    sigma = {b: int(sigma_k[b]) for b in range(k)}

    def mp(b: int) -> int:
        return sigma.get(int(b), int(b))
    after = 0.0
    for steps in executions:
        for i, st in enumerate(steps):
            st["local_ops"] = [([mp(q) for q in qs], U) for qs, U in st["local_ops"]]
            st["nonlocal_ops"] = [([mp(q) for q in qs], U) for qs, U in st["nonlocal_ops"]]
        # tiles: the needed bits move with their slots, the fill is chosen anew
        seg_first = [i for i, st in enumerate(steps) if "tile_needs" in st]
        for j, i in enumerate(seg_first):
            st = steps[i]
            end = seg_first[j + 1] if j + 1 < len(seg_first) else len(steps)
            slab = 0                          # local bits of the re-layout that closes this segment (already mapped)
            for later in steps[i:end]:
                for qs, U in later["nonlocal_ops"]:
                    if len(qs) == 2 and U.shape == (4, 4) and np.array_equal(U, _SWAP) and (qs[0] < k) != (qs[1] < k):
                        slab |= 1 << min(qs)
            new_masks, new_needs = [], []
            for pi, (mask, need) in enumerate(zip(st["tile_masks"], st["tile_needs"])):
                want = bin(int(mask)).count("1")
                tile = [mp(b) for b in bits_of(need)]
                forbid = slab if pi == len(st["tile_masks"]) - 1 else 0
                while len(tile) < want:
                    cands = [b for b in range(LINE_BITS, k) if b not in tile and not (forbid >> b) & 1]
                    if not cands:
                        cands = [b for b in range(LINE_BITS, k) if b not in tile]
                    tile.append(min(cands, key=lambda b: (tile_layout.tile_cost(model, tile + [b]), b)))
                after += tile_layout.tile_cost(model, tile)
                new_masks.append(sum(1 << b for b in tile))
                new_needs.append(sum(1 << mp(b) for b in bits_of(need)))
            st["tile_masks"], st["tile_needs"] = new_masks, new_needs
    return sigma, before, after
