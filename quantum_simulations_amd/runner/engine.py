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


class SingleGpuEngine:
    """The whole 2^n state on one MI355X (n <= 33 fits 288 GB)."""

    world = 1
    rank = 0

    def __init__(self, n_qubits: int, device: int = 0, mode: str = "fused"):
        self.n = n_qubits
        self.mode = mode
        self.state = DeviceChunk.empty(n_qubits, device)

    # ---- state ---------------------------------------------------------------------
    def init_zero_state(self) -> None:
        self.state.init_zero(True)

    def init_random_state(self, seed: int) -> None:
        self.state.init_random(seed)

    def norm2(self) -> float:
        return self.state.norm2()

    def state_vector(self) -> np.ndarray:
        return self.state.download()

    # ---- planning / execution --------------------------------------------------------
    def plan(self, circuit_dict: dict, repeats: int = 1) -> list:
        """Plan = list of passes, each a list of (qubits, U) handed to ONE C call (the same
        plan serves every repeat: the single-GPU layout never changes)."""
        cd = validate_circuit_dict(circuit_dict)
        if cd["number_of_qubits"] != self.n:
            raise ValueError(f"circuit has {cd['number_of_qubits']} qubits, engine has {self.n}")
        from quantum_simulations_amd.kernel.device import pack_ops
        if self.mode == "per-gate":
            return [pack_ops(gate_ops(cd))]
        return [pack_ops(p["local_ops"]) for p in batch_levels(levelize(cd), self.n)]

    def execute(self, plan: list) -> None:
        self.last_passes = 0
        for ops in plan:
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

    def sweep_1q(self, n: int, reps: int = 5) -> dict:
        """BASELINE config 3: H on every target of an n-qubit random state, per-target
        HIP-event timing, as fractions of the 8 TB/s HBM peak (32 * 2^n bytes per gate)."""
        dev = self.state if n == self.n else DeviceChunk.empty(n, self.state.device)
        dev.init_random(30)
        H = gate_table.H()
        per_target = []
        for q in range(n):
            dev.apply_1q(q, H)
            dev.sync()
            ts = []
            for _ in range(reps):
                dev.time_begin()
                dev.apply_1q(q, H)
                ts.append(dev.time_end())
            per_target.append(float(np.median(ts)))
        fr = [32.0 * (1 << n) / (ms * 1e-3) / 8.0e12 for ms in per_target]
        if dev is not self.state:
            dev.close()
        return {"n_qubits": n, "gate": "H", "ms_per_target": [round(t, 4) for t in per_target],
                "frac_of_8TBps": [round(f, 4) for f in fr], "min_frac": round(min(fr), 4),
                "median_frac": round(float(np.median(fr)), 4),
                "gate_apps_per_s_median": round(1e3 / float(np.median(per_target)), 1)}

    def prefix_parity(self, circuit_dict: dict, n_gates: int, expected: np.ndarray) -> float:
        """max |amp - expected| after the first n_gates gates from |0..0> (checker hook for
        bench.py's CPU leg; runs outside every timed region)."""
        cd = validate_circuit_dict(circuit_dict)
        self.init_zero_state()
        self.state.apply_ops(gate_ops({"gates": cd["gates"][:n_gates]}))
        worst = 0.0
        step = 1 << 24
        for off in range(0, 1 << self.n, step):
            cnt = min(step, (1 << self.n) - off)
            got = self.state.download(off, cnt)
            worst = max(worst, float(np.max(np.abs(got - expected[off:off + cnt]))))
        return worst

    def close(self) -> None:
        self.state.close()


def make_engine(n_qubits: int, world: int = 1, rank: int = 0, local_rank: int = 0,
                mode: str = "fused", **kw):
    if world == 1:
        return SingleGpuEngine(n_qubits, device=local_rank, mode=mode)
    from quantum_simulations_amd.runner.distributed import DistributedEngine
    return DistributedEngine(n_qubits, world, rank, local_rank, mode=mode, **kw)
