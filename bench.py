#!/usr/bin/env python3
"""Headline benchmark: gate-applications/sec (and achieved HBM GB/s) of the gate-application
hot path on MI355X.

    python bench.py --gpus 1 --steps 5 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): the seeded random 1q+CX circuit, depth 40, on
n = 28 + log2(N) qubits -- 2^28 complex128 amplitudes (4 GiB) per GPU, weak scaling.  One
"step" = one execution of the whole circuit on the HBM-resident state (the state is already
in HBM when the timed region starts; nothing crosses PCIe inside it).

`value` counts shard-level gate-applications per second summed over ranks (one gate of the
circuit applied to one rank's 2^28-amplitude shard = 1 unit; at N = 1 this is exactly
gate-applications/sec of the circuit).  `global_gate_apps_per_s` is the whole-state figure.
Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)
LOCAL_QUBITS = 28


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--local-qubits", type=int, default=LOCAL_QUBITS)
    ap.add_argument("--depth", type=int, default=40)
    ap.add_argument("--mode", choices=["fused", "per-gate"], default="fused",
                    help="fused: planner passes (batch_levels + tile fusion); per-gate: one launch per gate")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-sweep", action="store_true", help="skip the per-target H sweep (config 3)")
    ap.add_argument("--sweep-qubits", type=int, default=30)
    return ap.parse_args()


def host_core_share() -> int:
    """Cores this process may really use: min(cpu_count, affinity, cgroup cpu.max quota)."""
    cores = os.cpu_count() or 1
    try:
        cores = min(cores, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        quota, period = Path("/sys/fs/cgroup/cpu.max").read_text().split()
        if quota != "max":
            cores = min(cores, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return cores


def cpu_baseline(circuit: dict, budget_s: float) -> dict:
    """Time the C oracle (oracle/qsim_oracle.c, OpenMP over all host cores) on the first gates
    of the same circuit, on the same 2^n state size, for about `budget_s` seconds."""
    from oracle import c_oracle, dense_oracle
    n = circuit["number_of_qubits"]
    c_oracle.set_threads(host_core_share())
    psi = np.zeros(1 << n, dtype=np.complex128)
    psi[0] = 1.0
    done = 0
    t0 = time.perf_counter()
    for entry in circuit["gates"]:
        name, params, qubits = dense_oracle.decode_gate(entry)
        U = dense_oracle.gate_matrix(name, params)
        if len(qubits) == 1:
            c_oracle.apply_1q(psi, qubits[0], U)
        else:
            c_oracle.apply_2q(psi, qubits[0], qubits[1], U)
        done += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    return {"value": done / dt, "unit": "gate-applications/s", "cores": c_oracle.num_threads(),
            "kind": "port",
            "sample": f"first {done} gates of the same {n}-qubit circuit (full 2^{n} state, "
                      f"oracle/qsim_oracle.c with OpenMP, {dt:.1f} s)"}, psi, done


def cpu_v1_sql_baseline() -> dict:
    """BASELINE config 1: the v1 SQL engine's algorithm (oracle/v1_sql_oracle.py, stdlib sqlite3,
    one thread) on the 20-qubit GHZ circuit."""
    from oracle import v1_sql_oracle
    n = 20
    gates = [{"qubits": [0], "gate": "H"}] + [{"qubits": [q - 1, q], "gate": "CNOT"} for q in range(1, n)]
    t0 = time.perf_counter()
    psi = v1_sql_oracle.run_circuit({"number_of_qubits": n, "gates": gates})
    dt = time.perf_counter() - t0
    ok = bool(psi[0] == psi[-1] == 0.7071067811865475 and np.count_nonzero(psi) == 2)
    return {"value": n / dt, "unit": "gate-applications/s", "cores": 1, "kind": "port",
            "sample": f"20-qubit GHZ (20 gates), v1 SQL algorithm restated on stdlib sqlite3, {dt:.1f} s",
            "amplitudes_exact": ok}


def pmc_traffic_per_launch(kernel_prefix: str):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/*_pmc_summary.json, written by tools/pmc_summary.py); None if absent."""
    best = None
    for path in sorted((ROOT / "profiles").glob("*_pmc_summary.json")):
        try:
            doc = json.loads(path.read_text())
        except Exception:
            continue
        for row in doc.get("kernels", []):
            if row.get("kernel", "").startswith(kernel_prefix) and row.get("hbm_bytes_per_launch"):
                best = float(row["hbm_bytes_per_launch"])
    return best


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE {world}")
    if world & (world - 1):
        raise SystemExit("number of GPUs must be a power of two (shards are indexed by high qubits)")

    from quantum_simulations_amd.circuits import random_1q_cx_circuit
    from quantum_simulations_amd.runner.engine import make_engine

    k = args.local_qubits
    n = k + (world.bit_length() - 1)
    circuit = random_1q_cx_circuit(n, depth=args.depth)
    n_gates = len(circuit["gates"])

    engine = make_engine(n, world, rank, local_rank, mode=args.mode)
    engine.init_zero_state()
    plan = engine.plan(circuit, repeats=args.warmup + args.steps)

    for _ in range(args.warmup):
        engine.execute(plan)
    engine.barrier()
    if hasattr(engine, "reset_comm_stats"):
        engine.reset_comm_stats()
    engine.profile_begin()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        engine.execute(plan)
    engine.barrier()          # device sync on every rank + collective barrier
    dt_local = time.perf_counter() - t0
    prof = engine.profile_end()
    dt = engine.max_over_ranks(dt_local)
    norm2 = engine.norm2()

    if rank != 0:
        engine.close()
        return

    shard_gate_apps = n_gates * world * args.steps
    value = shard_gate_apps / dt
    dom = max(prof, key=lambda e: e["total_ms"]) if prof else None
    roofline = None
    if dom:
        secs = dom["total_ms"] * 1e-3
        achieved = dom["algorithmic_bytes"] / secs / 1e9
        moved = dom["hbm_bytes"] / secs / 1e9
        roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4),
                    "traffic": pmc_traffic_per_launch(dom["kernel"].split(" ")[0]),
                    "kernel": dom["kernel"], "launches": dom["launches"],
                    "avg_launch_ms": round(dom["total_ms"] / dom["launches"], 4),
                    "algorithmic_bytes_per_launch": dom["algorithmic_bytes"] / dom["launches"],
                    "hbm_bytes_moved_per_launch": dom["hbm_bytes"] / dom["launches"],
                    "hbm_GBps_moved": round(moved, 1), "hbm_frac_moved": round(moved / HBM_PEAK_GBS, 4),
                    "note": "achieved = algorithmic bytes (SURVEY 8d, summed over the gate-applications a "
                            "launch performs) / HIP-event time; a fused pass applies many gates per HBM "
                            "round trip, so achieved may exceed the physical peak; hbm_*_moved is what "
                            "the launch itself reads+writes (32 B per amplitude)"}
    total_moved = sum(e["hbm_bytes"] for e in prof)
    out = {
        "metric": "gate-applications/sec (random 1q+CX circuit, complex128 statevector)",
        "value": round(value, 2), "unit": "gate-applications/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{n}-qubit random 1q+CX circuit depth {args.depth} (seed 20260228), "
                               f"{n_gates} gates, complex128, {k} local qubits per GPU",
                   "n_qubits": n, "local_qubits": k, "gates_per_step": n_gates, "mode": args.mode,
                   "hbm_passes_per_step": engine.passes_per_step(plan),
                   "unit_note": "value = shard-level gate applications (gate x rank) per second"},
        "global_gate_apps_per_s": round(n_gates * args.steps / dt, 2),
        "hbm_GBps_moved_all_kernels_per_gpu": round(total_moved / dt / 1e9, 1),
        "norm2_after": norm2,
        "roofline": roofline,
        "kernel_breakdown": [{**e, "total_ms": round(e["total_ms"], 3)} for e in prof],
    }
    if world > 1:
        out["xgmi"] = engine.comm_stats()
    if not args.no_sweep and world == 1:
        # BASELINE config 3 / north-star target: H on every target of a 30-qubit state, per-gate
        # kernels (no fusion across the timed gates), fraction of the 8 TB/s peak per target
        out["sweep30"] = engine.sweep_1q(args.sweep_qubits)
        sw = out["sweep30"]
        out["roofline_per_gate_kernel"] = {
            "bound": "hbm", "kernel": "k_gate<2> / k_gate_shuffle<1,1>: dense 1q, one launch per gate",
            "achieved": round(sw["median_frac"] * HBM_PEAK_GBS, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": sw["median_frac"], "min_frac_over_targets": sw["min_frac"],
            "workload": f"H on every target of a {sw['n_qubits']}-qubit state (32 B x 2^n per launch)"}
    if not args.no_cpu_baseline and world == 1:
        base, psi_cpu, done = cpu_baseline(circuit, args.cpu_seconds)
        out["cpu_baseline"] = base
        # parity of the same prefix on the GPU (outside every timed region)
        out["parity_max_abs_diff_vs_cpu_prefix"] = engine.prefix_parity(circuit, done, psi_cpu)
        del psi_cpu
        out["cpu_baseline_v1_sql"] = cpu_v1_sql_baseline()
    engine.close()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
