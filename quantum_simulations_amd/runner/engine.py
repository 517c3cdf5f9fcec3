"""Execution engines: one HBM-resident state (single GPU) or one shard per rank (multi-GPU).

This is the collapsed form of the reference's runner layer (wenbo_engine/runner/
single_node.py:141-321): with the state resident in HBM a step is "launch kernels /
exchange / launch kernels" -- no chunk files, no double-buffer directories.

`make_engine(n, world, rank, local_rank)` returns an object with
    init_zero_state(), plan(circuit_dict) -> plan, execute(plan), barrier(), norm2(),
    state_vector() (logical order, gathered on every rank), close()
used by bench.py, the v3-style Driver and the tests.
"""
from __future__ import annotations

import numpy as np

from quantum_simulations_amd.circuit.fusion import batch_levels
from quantum_simulations_amd.circuit.io import levelize, validate_circuit_dict
from quantum_simulations_amd.kernel import gates as gate_table
from quantum_simulations_amd.kernel.device import DeviceChunk


def gate_ops(cd: dict) -> list:
    """Validated circuit -> [(qubits, U)] in list order (no levelling, no fusion)."""
    return [(g["qubits"], gate_table.gate_matrix(g["gate"], g["params"])) for g in cd["gates"]]


class GpuPlan(list):
    """A single-GPU plan: the list of batches ((nq, qubits, mats) packed for ONE C call each, in PHYSICAL index bits) plus
    `tiles` (per batch the high tile bits of its fused passes as uint64 masks, or None: let the library search) and `l2p`
    (the qubit layout the batches were written for: logical qubit q on index bit l2p[q]; None = identity)."""
    tiles: list
    l2p: list | None = None
    model_ms: tuple | None = None        # (identity, chosen) totals of the tile-cost model over the passes, when a layout was chosen


class SingleGpuEngine:
    """The whole 2^n state on one MI355X (n <= 33 fits 288 GB).

    layout = "auto" (fused mode, states of >= 26 qubits, plans made for >= 8 executions) / "search" (always): `plan`
    decides which index bit every logical qubit lives on --
      * the three qubits on the LINE bits 0..2 belong to every tile, so which three they are changes how many passes the
        greedy pass builder needs (17-20 for the 28-qubit bench circuit): a few dozen random choices are planned on the
        host in parallel (`qsim_plan_count_layouts`) and the cheapest kept;
      * the other qubits are placed so that the passes' tiles fall on index-bit sets with a good DRAM pattern
        (runner/tile_layout.py: a cost model fitted to measured passes, annealed by `qsim_choose_layout`) --
    and hands the SAME passes to the library on the new bits (`qsim_apply_ops_tiled`).  The state is then held in that
    layout -- the reference's `log_to_phys` notion (staging.py:587-658) -- and `state_vector()` / `logical_index()` undo
    it.  About a second of host work per plan, outside every timed region: worth it for plans that run many times.
    layout = "identity": index bit = qubit, no search."""

    world = 1
    rank = 0
    LAYOUT_MIN_QUBITS = 26
    LAYOUT_MIN_REPEATS = 8         # layout = "auto": plans for fewer executions are not worth a second of search
    LAYOUT_CANDIDATES = 384        # random relabellings (a quarter of them: only the three line-bit qubits) that are planned and counted
    LAYOUT_FINALISTS = 8           # minimum-pass layouts timed on the device (tune_on_device, |0..0> state only)

    def __init__(self, n_qubits: int, device: int = 0, mode: str = "fused", layout: str = "auto", tune_on_device: bool = False):
        if layout not in ("auto", "search", "identity"):
            raise ValueError("layout must be 'auto', 'search' or 'identity'")
        self.n = n_qubits
        self.mode = mode
        self.layout_mode = layout
        self.tune_on_device = bool(tune_on_device)   # plan(): time the minimum-pass layouts on the device when the state is |0..0>
        self.state = DeviceChunk.empty(n_qubits, device)
        self.l2p: list | None = None        # layout of the state right now (None = identity)
        self._zero = False                  # the state is |0..0>: the same vector in every layout

    # ---- state ---------------------------------------------------------------------
    def init_zero_state(self) -> None:
        self.state.init_zero(True)
        self.l2p, self._zero = None, True

    def init_random_state(self, seed: int) -> None:
        self.state.init_random(seed)
        self.l2p, self._zero = None, False

    def norm2(self) -> float:
        return self.state.norm2()

    def state_vector(self) -> np.ndarray:
        """The state in LOGICAL qubit order (the layout undone on the host: small n)."""
        psi = self.state.download()
        if self.l2p is None:
            return psi
        from quantum_simulations_amd.circuit.staging import permute_state
        return permute_state(psi, self.l2p)

    def logical_index(self, offset: int, count: int) -> np.ndarray:
        """Logical amplitude indices of the physical range [offset, offset + count) of the state in its current layout."""
        x = offset + np.arange(count, dtype=np.int64)
        if self.l2p is None:
            return x
        y = np.zeros_like(x)
        for q, p in enumerate(self.l2p):
            y |= ((x >> p) & 1) << q
        return y

    # ---- planning / execution --------------------------------------------------------
    def plan(self, circuit_dict: dict, repeats: int = 1) -> GpuPlan:
        """Plan = list of batches, each handed to ONE C call (the same plan serves every repeat: the layout of a
        single-GPU run is chosen once per plan and never changes)."""
        cd = validate_circuit_dict(circuit_dict)
        if cd["number_of_qubits"] != self.n:
            raise ValueError(f"circuit has {cd['number_of_qubits']} qubits, engine has {self.n}")
        from quantum_simulations_amd.kernel.device import pack_ops
        if self.mode == "per-gate":
            plan = GpuPlan([pack_ops(gate_ops(cd))])
            plan.tiles = [None]
            return plan
        batches = [p["local_ops"] for p in batch_levels(levelize(cd), self.n)]
        search = self.layout_mode == "search" or (self.layout_mode == "auto" and repeats >= self.LAYOUT_MIN_REPEATS)
        if not search or self.n < self.LAYOUT_MIN_QUBITS:
            plan = GpuPlan([pack_ops(ops) for ops in batches])
            plan.tiles = [None] * len(batches)
            return plan
        import time
        t0 = time.perf_counter()
        tune = self.tune_on_device and self._zero            # (timing runs overwrite the state: only |0..0> can be put back)
        finalists = choose_plan_layout(self.n, batches, self.LAYOUT_CANDIDATES, n_finalists=self.LAYOUT_FINALISTS if tune else 4,
                                       all_finalists=True)

        def build(l2p, masks, info):
            plan = GpuPlan([pack_ops([([l2p[q] for q in qs], U) for qs, U in ops]) for ops in batches])
            plan.tiles = masks
            plan.l2p = None if l2p == list(range(self.n)) else l2p
            plan.model_ms = info["model_ms"]
            plan.layout_info = dict(info)
            return plan
        plans = [build(*f) for f in finalists]
        chosen = plans[0]
        if tune and len(plans) > 1:
            # the cost model cannot rank the minimum-pass layouts (measured: 3 % apart, uncorrelated with the model): the
            # device can -- two timed executions of each on the (zero) state, which is |0..0> again afterwards
            timed = []
            for p in plans:
                self.init_zero_state()
                self.execute(p)
                ms = []
                for _ in range(2):
                    self.state.time_begin()
                    self.execute(p)
                    ms.append(self.state.time_end())
                timed.append(min(ms))
            chosen = plans[int(np.argmin(timed))]
            chosen.layout_info["tuned_on_device_ms"] = [round(t, 3) for t in timed]
            self.init_zero_state()
        chosen.layout_info["seconds"] = round(time.perf_counter() - t0, 3)
        return chosen

    def _adopt_layout(self, l2p) -> None:
        """Bring the state into the layout a plan was written for."""
        if l2p == self.l2p or (l2p is None and self.l2p == list(range(self.n))):
            return
        if self._zero:                      # |0..0> looks the same in every layout
            self.l2p = l2p
            return
        # a non-trivial state: move the qubits with SWAP gates (fused passes; rare -- a new plan on a used state)
        SW = gate_table.SWAP()
        swaps = [([a, b], SW) for a, b in layout_swaps(self.l2p, l2p, self.n)]
        if swaps:
            self.state.apply_ops(swaps)
        self.l2p = l2p

    def execute(self, plan) -> None:
        self._adopt_layout(getattr(plan, "l2p", None))
        self._zero = False
        self.last_passes = 0
        tiles = getattr(plan, "tiles", None) or [None] * len(plan)
        for ops, masks in zip(plan, tiles):
            if masks is not None and self.mode != "per-gate":
                self.last_passes += self.state.apply_ops_tiled(ops, masks)
            else:
                self.last_passes += self.state.apply_ops(ops, fused=self.mode != "per-gate")

    def passes_per_step(self, plan: list) -> int:
        """HBM round trips of the last executed step (fused tile launches or single gates)."""
        return getattr(self, "last_passes", sum(len(ops[0]) for ops in plan))

    # ---- synchronisation / measurement --------------------------------------------------
    def barrier(self) -> None:
        self.state.sync()

    def max_over_ranks(self, value: float) -> float:
        return value

    def profile_begin(self) -> None:
        self.state.profile_begin()

    def profile_end(self) -> list[dict]:
        return self.state.profile_end()

    def sweep_gates(self, n: int, reps: int = 5) -> dict:
        """BASELINE config 3: one gate per launch on an n-qubit random state, per-target HIP-event
        timing (median of `reps`), as fractions of the 8 TB/s HBM peak of SURVEY 8d's algorithmic bytes:
        H(q) 32 * 2^n for every q; secondary rows T(q), CNOT(q, q+1), CNOT(0, q) at 16 * 2^n."""
        dev = self.state if n == self.n else DeviceChunk.empty(n, self.state.device)
        dev.init_random(30)
        if dev is self.state:
            self.l2p, self._zero = None, False
        H, T, CX = gate_table.H(), gate_table.T(), gate_table.CNOT()
        N = 1 << n

        def timed(fn) -> float:
            fn()
            dev.sync()
            ts = []
            for _ in range(reps):
                dev.time_begin()
                fn()
                ts.append(dev.time_end())
            return float(np.median(ts))

        def moved_bytes(fn) -> float:
            """bytes the launch(es) of one gate really move (the library's own per-launch accounting: 32 B per amplitude
            a launch touches -- all of them when a diagonal / control bit lies inside a 128-B line)"""
            dev.profile_begin()
            fn()
            return float(sum(e["hbm_bytes"] for e in dev.profile_end()))

        def row(label, targets, fn, nbytes, note=None) -> dict:
            ms = [timed(lambda q=q: fn(q)) for q in targets]
            fr = [nbytes / (t * 1e-3) / 8.0e12 for t in ms]
            moved = [moved_bytes(lambda q=q: fn(q)) for q in targets]
            mf = [b / (t * 1e-3) / 8.0e12 for b, t in zip(moved, ms)]
            out = {"targets": [int(q) for q in targets], "algorithmic_bytes_per_gate": nbytes,
                   "ms_per_target": [round(t, 4) for t in ms], "frac_of_8TBps": [round(f, 4) for f in fr],
                   # the same launches by the bytes they MOVE: where it exceeds frac_of_8TBps (sub-line control / diagonal
                   # bits: every 128-B line moves for half or a quarter of the amplitudes) the algorithmic fraction is
                   # at its floor, not the kernel slow
                   "moved_bytes_per_target": [int(b) for b in moved], "moved_frac": [round(f, 4) for f in mf],
                   "min_frac": round(min(fr), 4), "median_frac": round(float(np.median(fr)), 4),
                   "max_frac": round(max(fr), 4), "min_moved_frac": round(min(mf), 4),
                   "gate_apps_per_s_median": round(1e3 / float(np.median(ms)), 1)}
            if note:
                out["note"] = note
            return out

        subline = ("a diagonal / control bit below index bit 3 shares every 128-B line with the untouched half: every "
                   "line must move, so the algorithmic fraction of those targets is capped near 0.37 by construction")
        rows = {"H(q)": row("H", range(n), lambda q: dev.apply_1q(q, H), 32 * N),
                "T(q)": row("T", range(n), lambda q: dev.apply_1q(q, T), 16 * N, subline),
                "CNOT(q,q+1)": row("CX", range(n - 1), lambda q: dev.apply_2q(q, q + 1, CX), 16 * N, subline),
                "CNOT(0,q)": row("CX0", range(1, n), lambda q: dev.apply_2q(0, q, CX), 16 * N, subline)}
        # the one GEMM-shaped op of the path: dense 2^k x 2^k blocks on the matrix cores (qsim_apply_fused_k -> k_dense_mfma),
        # random unitaries on low / middle / high / mixed index bits above the 128-byte line
        rng = np.random.default_rng(4)
        dense = {}
        MFMA_F64_PEAK_TFLOPS = 78.6     # MI355X dense fp64 matrix peak (MI355X_MICROARCH.md)
        for k in (3, 4, 5, 6):
            mixed = [5, 14, n - 2, n - 9, 8, 12][:k]
            sets = [list(range(3, 3 + k)), list(range(10, 10 + k)), list(range(n - k, n)), mixed, list(range(k))]
            sets = [qs for qs in sets if max(qs) < n and len(set(qs)) == k]
            M = np.linalg.qr(rng.standard_normal((1 << k, 1 << k)) + 1j * rng.standard_normal((1 << k, 1 << k)))[0]
            ms = [timed(lambda qs=qs: dev.apply_fused_k(qs, M)) for qs in sets]
            fr = [32 * N / (t * 1e-3) / 8.0e12 for t in ms]
            flop = 8.0 * (1 << k) * N                         # 2^k complex multiply-adds per amplitude
            tf = [flop / (t * 1e-3) / 1e12 for t in ms]
            # which unit bounds the kernel at this size: the matrix cores when the flops at their peak take longer than the
            # bytes at the HBM peak (k = 6: 512 flop per amplitude = 7.0 ms against 4.3 ms at 30 qubits)
            bound = "mfma" if flop / (MFMA_F64_PEAK_TFLOPS * 1e12) > 32.0 * N / 8.0e12 else "hbm"
            dense[f"k={k}"] = {"qubit_sets": sets, "sets_are": ["low", "middle", "high", "mixed", "line bits 0.." + str(k - 1)][:len(sets)],
                               "ms": [round(t, 4) for t in ms], "frac_of_8TBps": [round(f, 4) for f in fr],
                               "flop_per_amplitude": 8 * (1 << k), "achieved_TFLOPs": [round(x, 2) for x in tf],
                               "frac_of_mfma_f64_peak": [round(x / MFMA_F64_PEAK_TFLOPS, 4) for x in tf], "bound": bound,
                               "kernel": f"k_dense_mfma2<{k}> (v_mfma_f64_16x16x4_f64, 16 blocks per wave and step, in place; "
                                         + ("matrix image in LDS" if k >= 5 else "matrix image in registers") + ")"}
        norm2 = dev.norm2()
        if dev is not self.state:
            dev.close()
        return {"n_qubits": n, "reps": reps, "rows": rows, "dense_blocks": dense, "norm2_after": norm2}

    def copy_ceiling(self, reps: int = 7) -> dict:
        """Same-run device-to-device copy of a buffer of the state's size (qsim_copy: streaming loads and
        stores, 32 B moved per amplitude): the practical HBM ceiling next to the nominal 8 TB/s."""
        other = DeviceChunk.empty(self.n, self.state.device)
        other.copy_from(self.state)
        other.sync()
        ts = []
        for _ in range(reps):
            other.time_begin()
            other.copy_from(self.state)
            ts.append(other.time_end())
        other.close()
        ms = float(np.median(ts))
        gbps = 32.0 * (1 << self.n) / (ms * 1e-3) / 1e9
        return {"GBps": round(gbps, 1), "frac_of_8TBps": round(gbps / 8000.0, 4), "ms": round(ms, 4),
                "bytes": 32 * (1 << self.n), "kernel": "k_copy (read + write, non-temporal above 256 MiB)"}

    def stream_ceiling(self, reps: int = 5) -> dict:
        """What this device streams at the state's size in THIS run: the best of the copy kernel (non-temporal and plain),
        the runtime's device-to-device copy, and an in-place read-modify-write of every amplitude (H on a middle index
        bit: one 32-B round trip per amplitude, the access pattern of the gate kernels themselves).  All move 32 B per
        amplitude.  bench.py reports the fused pass against the best of them (`frac_of_achievable`): unlike the copy
        kernel alone (r04: 0.71 of peak while the in-place H reached 0.77 in the same run) the maximum IS a ceiling of
        what was seen to stream on this box."""
        other = DeviceChunk.empty(self.n, self.state.device)
        nbytes = 32.0 * (1 << self.n)
        H = gate_table.H()

        def timed(fn) -> float:
            fn()
            other.sync()
            ts = []
            for _ in range(reps):
                other.time_begin()
                fn()
                ts.append(other.time_end())
            return float(np.median(ts))
        other.copy_from(self.state)
        rows = {"copy_nontemporal": timed(lambda: other.copy_from(self.state, 1)),
                "copy_plain": timed(lambda: other.copy_from(self.state, 2)),
                "copy_hipMemcpyAsync": timed(lambda: other.copy_from(self.state, 3)),
                "inplace_rmw_H_bit12": timed(lambda: other.apply_1q(min(12, self.n - 1), H)),
                "inplace_rmw_H_bit5": timed(lambda: other.apply_1q(min(5, self.n - 1), H))}
        other.close()
        gbps = {name: round(nbytes / (ms * 1e-3) / 1e9, 1) for name, ms in rows.items()}
        best = max(gbps, key=gbps.get)
        return {"GBps": gbps[best], "best": best, "frac_of_8TBps": round(gbps[best] / 8000.0, 4), "candidates_GBps": gbps,
                "bytes": int(nbytes), "note": "each candidate moves 32 B per amplitude of a buffer of the state's size; median of %d" % reps}

    def prefix_parity(self, circuit_dict: dict, n_gates: int, expected: np.ndarray) -> float:
        """max |amp - expected| after the first n_gates gates from |0..0> (checker hook for bench.py's CPU leg; runs
        outside every timed region) -- through the SAME path as the timed steps: planned by `plan` (layout search and named
        tiles included, when the engine uses them), executed, compared through the layout."""
        cd = validate_circuit_dict(circuit_dict)
        prefix = {"number_of_qubits": self.n, "gates": cd["gates"][:n_gates]}
        self.init_zero_state()
        tune, self.tune_on_device = self.tune_on_device, False          # (no timing runs for a check)
        try:
            plan = self.plan(prefix, repeats=self.LAYOUT_MIN_REPEATS)
        finally:
            self.tune_on_device = tune
        self.execute(plan)
        worst = 0.0
        step = 1 << 24
        for off in range(0, 1 << self.n, step):
            cnt = min(step, (1 << self.n) - off)
            got = self.state.download(off, cnt)
            want = expected[off:off + cnt] if self.l2p is None else expected[self.logical_index(off, cnt)]
            worst = max(worst, float(np.max(np.abs(got - want))))
        self.last_parity_layout = None if self.l2p is None else list(self.l2p)
        return worst

    def close(self) -> None:
        self.state.close()


def layout_swaps(cur, want, n: int) -> list:
    """Index-bit pairs whose SWAPs, applied in order, take a state from layout `cur` to layout `want` (logical qubit q on
    index bit layout[q]; None = identity): at most n - 1 of them."""
    cur = list(cur) if cur is not None else list(range(n))
    want = list(want) if want is not None else list(range(n))
    at = {p: q for q, p in enumerate(cur)}              # index bit -> the logical qubit on it
    swaps = []
    for q in range(n):
        a, b = cur[q], want[q]
        if a != b:
            other = at[b]                               # the qubit sitting where q has to go
            swaps.append((a, b))
            at[a], at[b] = other, q
            cur[other], cur[q] = a, b
    return swaps


def _count_passes(n: int, batches, layouts: np.ndarray, threads: int) -> np.ndarray:
    """Fused passes of the batches under each layout (rows of `layouts`: qubit -> index bit), planned on the host."""
    import ctypes as C

    from quantum_simulations_amd import _lib
    from quantum_simulations_amd.kernel.device import pack_ops
    lib = _lib.load()
    total = np.zeros(len(layouts), dtype=np.int64)
    lay = np.ascontiguousarray(layouts, dtype=np.int32)
    for ops in batches:
        nq, qubits, mats = pack_ops(ops)
        if len(nq) < 2:
            total += len(nq)
            continue
        out = np.zeros(len(lay), dtype=np.int32)
        _lib.check(lib.qsim_plan_count_layouts(n, len(nq), nq.ctypes.data_as(C.c_void_p), qubits.ctypes.data_as(C.c_void_p),
                                               mats.ctypes.data_as(C.c_void_p), len(lay), lay.ctypes.data_as(C.c_void_p),
                                               out.ctypes.data_as(C.c_void_p), threads))
        total += out
    return total


def choose_plan_layout(n: int, batches, n_candidates: int = 384, seed: int = 20260504, n_finalists: int = 4,
                       all_finalists: bool = False):
    """(l2p, tile masks per batch on the chosen index bits, info) for the op lists `batches` (logical qubits): the line-bit
    qubits that need the fewest passes among `n_candidates` random choices (the identity included), then the other
    qubits placed by the tile-cost model (runner/tile_layout.py).  Host only.  all_finalists: the list of up to
    n_finalists minimum-pass layouts, best model cost first (the engine may time them on the device)."""
    import os

    from quantum_simulations_amd.runner import tile_layout
    rng = np.random.default_rng(seed)
    layouts = np.tile(np.arange(n, dtype=np.int32), (n_candidates + 1, 1))
    for i, row in enumerate(layouts[1:]):
        if i % 4 == 0:                                      # a quarter: only the three line-bit qubits change
            for bit, q in enumerate(int(x) for x in rng.choice(n, size=3, replace=False)):
                j = int(np.flatnonzero(row == bit)[0])      # the qubit on `bit` trades places with q
                row[j], row[q] = row[q], bit
        else:                                               # the rest: every qubit somewhere else (the greedy builder's
            row[:] = rng.permutation(n)                     # tie-breaks walk the bits in order: other labels, other plans)
    try:
        threads = len(os.sched_getaffinity(0))
    except AttributeError:
        threads = os.cpu_count() or 1
    counts = _count_passes(n, batches, layouts, max(1, min(16, threads)))
    best = int(counts.min())
    finalists = [int(i) for i in np.flatnonzero(counts == best)][:max(1, n_finalists)]   # (the identity first when it ties)
    out = []
    for f in finalists:
        first = [int(x) for x in layouts[f]]
        moved = [[([first[q] for q in qs], U) for qs, U in ops] for ops in batches]
        masks = [_planned_tile_masks(n, ops) for ops in moved]
        tiles = [[b for b in range(tile_layout.LOW, n) if (int(m) >> b) & 1] for ms in masks for m in ms]
        second, cost0, cost1 = min((tile_layout.choose_layout(tiles, n, seed=s) for s in range(1, 9)), key=lambda r: r[2])
        l2p = [second[first[q]] for q in range(n)]
        final_masks = [np.array([sum(1 << second[b] for b in range(n) if (int(m) >> b) & 1) for m in ms], dtype=np.uint64) for ms in masks]
        info = {"passes_identity": int(counts[0]), "passes_chosen": best, "candidates": n_candidates + 1,
                "candidates_by_passes": {int(c): int((counts == c).sum()) for c in np.unique(counts)},
                "model_ms": (round(cost0, 3), round(cost1, 3))}
        out.append((l2p, final_masks, info))
    out.sort(key=lambda r: r[2]["model_ms"][1])
    return out if all_finalists else out[0]


def _planned_tile_masks(n: int, ops) -> np.ndarray:
    """The high tile bits of the fused passes the library plans for `ops` on n qubits, one uint64 mask per pass
    (qsim_plan_ops: the host planner without a device; offset of `h` in a pass image: csrc/tile_kernel.h TileArgs)."""
    import ctypes as C

    from quantum_simulations_amd import _lib
    from quantum_simulations_amd.kernel.device import pack_ops
    nq, qubits, mats = pack_ops(ops)
    lib = _lib.load()
    count = C.c_int32()
    if len(nq) < 2:
        return np.zeros(0, dtype=np.uint64)
    args = (n, len(nq), nq.ctypes.data_as(C.c_void_p), qubits.ctypes.data_as(C.c_void_p), mats.ctypes.data_as(C.c_void_p))
    _lib.check(lib.qsim_plan_ops(*args, None, 0, C.byref(count)))
    images = np.zeros((count.value, 4096), dtype=np.uint8)
    _lib.check(lib.qsim_plan_ops(*args, images.ctypes.data_as(C.c_void_p), images.nbytes, C.byref(count)))
    out = np.zeros(count.value, dtype=np.uint64)
    for p in range(count.value):
        T = int(images[p, 12:16].view("<i4")[0])
        out[p] = sum(1 << int(b) for b in images[p, 16:16 + T - 3])
    return out


def make_engine(n_qubits: int, world: int = 1, rank: int = 0, local_rank: int = 0,
                mode: str = "fused", **kw):
    if world == 1:                      # (rehearsal / exchange only mean something with more than one rank)
        return SingleGpuEngine(n_qubits, device=local_rank, mode=mode, layout=kw.get("layout", "auto"),
                               tune_on_device=kw.get("tune_on_device", False))
    from quantum_simulations_amd.runner.distributed import DistributedEngine
    return DistributedEngine(n_qubits, world, rank, local_rank, mode=mode, **kw)
