#!/usr/bin/env python3
"""Headline benchmark: gate-applications/sec and achieved HBM GB/s of the gate-application hot
path on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python bench.py --gpus N --steps K --warmup W          # N > 1 without a launcher: this process starts its N ranks
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N --dry-run                     # comm schedule only (gloo, no shards, no GPU)
    python bench.py --gpus 4 --rehearsal --local-qubits 24 # N ranks sharing the visible GPU(s), host-staged gloo exchange

Started without WORLD_SIZE in the environment, `--gpus N > 1` makes this process the PARENT of N fresh rank processes
(one per GPU, 127.0.0.1 rendezvous on a free port -- the per-chunk task fan-out of the reference's
runner/spark_runner.py:128-136 with one long-lived task per shard).  The parent never touches the GPU, relays rank 0's
single JSON line, and exits non-zero if any rank does, ending the others instead of waiting for them.

Workload: the seeded random 1q+CX circuit, depth 40 (BASELINE configs[1]).  N = 1: 28 qubits (4 GiB,
the configuration the metric is quoted on).  N > 1: 30 LOCAL qubits per GPU (16 GiB shards, weak
scaling: n = 30 + log2 N, so --gpus 8 is the 33-qubit run the north star names).  One "step" = one
execution of the whole circuit on the HBM-resident state; nothing crosses PCIe in the timed region.

`value` = gate-applications per second of the WHOLE 2^n state at every N (one gate of the
circuit applied to the full state = 1).  Rank 0 prints ONE JSON line and exits non-zero when the run
is invalid (parity or norm check failed, probe knobs in the environment).
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

T0 = time.time()       # process start: the wall-clock budget of an N > 1 run counts from here
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)
PARITY_TOL = 1e-10
NORM_TOL = 1e-9
ALLOWED_ENV: set = set()                 # no QSIM_* variable is read by anything this script times


class SectionBudget:
    """Wall-clock bookkeeping of an N > 1 run (the driver gives one command a fixed time limit): every section's elapsed
    seconds go to stderr as it starts, OPTIONAL sections are skipped -- and named in the line -- once `limit_s` seconds of
    the process are gone.  `agree`: a collective max over the ranks, so that every rank takes the same decision (a section
    that some ranks enter and others skip would hang in its first collective)."""

    def __init__(self, limit_s: float, t0: float | None = None, clock=time.time, agree=None, log=None):
        self.limit_s, self.t0, self.clock, self.agree = float(limit_s), (clock() if t0 is None else t0), clock, agree
        self.log = log or (lambda msg: print(msg, file=sys.stderr, flush=True))
        self.sections, self.skipped, self._open = [], [], None

    def elapsed(self) -> float:
        e = self.clock() - self.t0
        return float(self.agree(e)) if self.agree is not None else e

    def begin(self, name: str, optional: bool = False) -> bool:
        now = self.elapsed()
        self.end(now)
        if optional and now > self.limit_s:
            self.skipped.append(name)
            self.log(f"bench.py: [{now:6.1f} s] SKIPPED {name} (the {self.limit_s:.0f} s budget is spent)")
            return False
        self.log(f"bench.py: [{now:6.1f} s] {name}")
        self._open = (name, now)
        return True

    def end(self, now: float | None = None) -> None:
        """(no collective here: `report` runs on rank 0 alone)"""
        if self._open is not None:
            now = (self.clock() - self.t0) if now is None else now
            self.sections.append({"name": self._open[0], "seconds": round(now - self._open[1], 2)})
            self._open = None

    def report(self) -> dict:
        self.end()
        return {"limit_s": self.limit_s, "sections": self.sections, "skipped": self.skipped,
                "total_s": round(self.clock() - self.t0, 1)}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--local-qubits", type=int, default=0, help="default: 28 at N = 1, 30 at N > 1")
    ap.add_argument("--depth", type=int, default=40)
    ap.add_argument("--mode", choices=["fused", "per-gate"], default="fused",
                    help="fused: planner passes (batch_levels + tile fusion); per-gate: one launch per gate")
    ap.add_argument("--layout", choices=["auto", "search", "identity"], default="auto",
                    help="N = 1: auto / search = the engine chooses which index bit a qubit lives on (line-bit qubits for the "
                         "pass count, the rest by the tile-pattern model); identity = bit q")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-sweep", action="store_true", help="skip the per-target sweeps (config 3)")
    ap.add_argument("--no-api-path", action="store_true", help="N = 1: skip the one-call timings of single_node.run / Driver.run_circuit")
    ap.add_argument("--no-plan-check", action="store_true",
                    help="N = 1: skip the timed plan's fingerprint check against an identity-layout execution (profiling runs: "
                         "keeps other plans' passes out of the kernel statistics)")
    ap.add_argument("--sweep-qubits", type=int, default=30)
    ap.add_argument("--fused-qubits", type=int, default=30,
                    help="N = 1: size of the second fused run (`fused30` sub-record; 0 = off)")
    ap.add_argument("--sustain-seconds", type=float, default=3.0,
                    help="second, longer timed region after the K-step one (0 = off)")
    ap.add_argument("--no-configs", action="store_true", help="N > 1: skip the config-4 / config-5 sub-runs")
    ap.add_argument("--no-single-reference", action="store_true",
                    help="N > 1: skip the one-GPU run at the same local size on rank 0")
    ap.add_argument("--dry-run", action="store_true",
                    help="N > 1: print and cross-check the communication schedule only (gloo, no shard memory)")
    ap.add_argument("--rehearsal", action="store_true",
                    help="N > 1: the ranks share the visible GPU(s) and exchange through host-staged gloo (RCCL refuses two "
                         "ranks on one device); the line says \"exchange\": \"gloo-rehearsal\" and is not a multi-GPU number")
    ap.add_argument("--exchange", choices=["torch", "cabi"], default="torch",
                    help="N > 1: who posts the RCCL transfers: torch.distributed P2P (default) or the library's own "
                         "communicator (qsim_comm_exchange, C ABI)")
    ap.add_argument("--no-amplitude-check", action="store_true",
                    help="N > 1: skip the one-GPU reference runs behind the per-shard fingerprint comparison")
    ap.add_argument("--budget-seconds", type=float, default=400.0,
                    help="N > 1: optional sections (re-layout measurements, fused on / off, the other exchange API, the one-GPU "
                         "run at the same local size) are skipped, and named in the line, once this much wall time is gone")
    ap.add_argument("--no-relayout-pipeline", action="store_true",
                    help="N > 1: plain fused re-layouts (whole slabs, one group, waited for before the shard is read): isolates "
                         "an ordering problem of the piece pipeline from a wrong schedule")
    ap.add_argument("--other-api-timeout", type=float, default=150.0,
                    help="N > 1: seconds the second exchange API's section may take before the line is printed without it and the ranks exit")
    ap.add_argument("--ab-steps", type=int, default=3, help="N > 1: steps of the short A/B regions (fused re-layout off, other exchange API)")
    return ap.parse_args()


def refuse_probe_environment():
    """A QSIM_* variable changes planning (or, in the probe build, results): a number timed under one
    is not the product's number."""
    bad = sorted(k for k in os.environ if k.startswith("QSIM_") and k not in ALLOWED_ENV)
    if bad:
        print(json.dumps({"error": "refusing to run with tuning / probe knobs in the environment", "knobs": bad}),
              flush=True)
        sys.exit(2)


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


# ---------------------------------------------------------------------------------- CPU columns
def cpu_baseline(circuit: dict, budget_s: float):
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


def cpu_config1_columns() -> dict:
    """BASELINE config 1 (20-qubit GHZ: H + CNOT ladder, complex128, single process) on this node's
    host cores, three ways: the v1 SQL engine's algorithm on stdlib sqlite3 (1 core),
    the numpy restatement of cpu_scalar.apply_1q / apply_2q (1 thread), the C loop with OpenMP."""
    from oracle import c_oracle, dense_oracle, v1_sql_oracle
    n = 20
    gates = [{"qubits": [0], "gate": "H"}] + [{"qubits": [q - 1, q], "gate": "CNOT"} for q in range(1, n)]
    cd = {"number_of_qubits": n, "gates": gates}
    want0 = 0.7071067811865475

    def exact(psi) -> bool:
        return bool(abs(psi[0] - want0) < 1e-15 and abs(psi[-1] - want0) < 1e-15 and np.count_nonzero(psi) == 2)

    out = {}
    t0 = time.perf_counter()
    psi = v1_sql_oracle.run_circuit(cd)
    dt = time.perf_counter() - t0
    out["v1_sql_sqlite3"] = {"value": n / dt, "unit": "gate-applications/s", "cores": 1, "kind": "port",
                             "sample": f"20-qubit GHZ (20 gates), v1 SQL algorithm restated on stdlib sqlite3, {dt:.2f} s",
                             "amplitudes_exact": exact(psi)}
    t0 = time.perf_counter()
    psi = dense_oracle.simulate(cd)
    dt = time.perf_counter() - t0
    out["numpy_dense_1_thread"] = {"value": n / dt, "unit": "gate-applications/s", "cores": 1, "kind": "port",
                                   "sample": f"20-qubit GHZ, oracle/dense_oracle.py (restatement of cpu_scalar.apply_1q/2q, "
                                             f"pinned by golden G3), {dt:.2f} s",
                                   "amplitudes_exact": exact(psi)}
    c_oracle.set_threads(host_core_share())
    c_oracle.simulate(cd)                                    # warm the thread pool
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        psi = c_oracle.simulate(cd)
    dt = (time.perf_counter() - t0) / reps
    out["c_openmp"] = {"value": n / dt, "unit": "gate-applications/s", "cores": c_oracle.num_threads(), "kind": "port",
                       "sample": f"20-qubit GHZ, oracle/qsim_oracle.c with OpenMP, {dt * 1e3:.1f} ms per run",
                       "amplitudes_exact": exact(psi)}
    return out


def pmc_traffic_per_launch(kernel_prefix: str, local_qubits: int = 28):
    """(HBM bytes per launch of the dominant kernel, the summary it comes from) out of the committed rocprofv3 PMC
    passes (profiles/*_pmc_summary.json, written by tools/pmc_summary.py) -- but only from a summary taken with
    the kernel sources this run was built from (`csrc_sha16` == _lib.source_hash()); anything older is a stale
    figure and is reported as null.  The PMC passes cannot run inside this process (rocprofv3 wraps it)."""
    from quantum_simulations_amd._lib import source_hash
    now = source_hash()
    best = (None, None)
    for path in sorted((ROOT / "profiles").glob("*_pmc_summary.json")):
        try:
            doc = json.loads(path.read_text())
        except Exception:
            continue
        if doc.get("csrc_sha16") != now or doc.get("local_qubits", 28) != local_qubits:
            continue
        for row in doc.get("kernels", []):
            if row.get("kernel", "").startswith(kernel_prefix) and row.get("hbm_bytes_per_launch"):
                best = (float(row["hbm_bytes_per_launch"]), f"profiles/{path.name}")
    return best


def fused_run(n: int, depth: int, steps: int, warmup: int, device: int, layout: str = "auto") -> dict:
    """The fused workload at another size on a fresh state (N = 1: the 30-qubit random 1q+CX circuit, the largest
    single-GPU configuration of BASELINE.json's list that bench.py times): passes, ms per pass by HIP events,
    roofline fraction of the bytes the launches move, gate-applications/s by the host clock."""
    from quantum_simulations_amd.circuits import random_1q_cx_circuit
    from quantum_simulations_amd.runner.engine import SingleGpuEngine
    # (a few steps of a run that would repeat the plan many times: the layout search is made as for such a run)
    eng = SingleGpuEngine(n, device=device, mode="fused", layout="search" if layout == "auto" else layout,
                          tune_on_device=layout == "auto")
    circuit = random_1q_cx_circuit(n, depth=depth)
    n_gates = len(circuit["gates"])
    eng.init_zero_state()
    plan = eng.plan(circuit, repeats=warmup + steps)
    for _ in range(warmup):
        eng.execute(plan)
    eng.barrier()
    eng.profile_begin()
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.execute(plan)
    eng.barrier()
    dt = time.perf_counter() - t0
    prof = eng.profile_end()
    norm2 = eng.norm2()
    passes = eng.passes_per_step(plan)
    eng.close()
    dom = max(prof, key=lambda e: e["total_ms"])
    moved = dom["hbm_bytes"] / (dom["total_ms"] * 1e-3) / 1e9
    return {"workload": f"{n}-qubit random 1q+CX circuit depth {depth} (seed 20260228), {n_gates} gates, complex128, fused",
            "n_qubits": n, "gates_per_step": n_gates, "steps": steps, "warmup": warmup,
            "gate_apps_per_s": round(n_gates * steps / dt, 2), "ms_per_step": round(dt / steps * 1e3, 3),
            "hbm_passes_per_step": passes, "kernel": dom["kernel"], "launches": dom["launches"],
            "avg_launch_ms": round(dom["total_ms"] / dom["launches"], 4),
            "bytes_per_launch": dom["hbm_bytes"] / dom["launches"],
            "achieved_GBps": round(moved, 1), "frac": round(moved / HBM_PEAK_GBS, 4), "norm2_after": norm2,
            "qubit_layout": "identity" if getattr(plan, "l2p", None) is None else
                            {"logical_to_index_bit": plan.l2p, **getattr(plan, "layout_info", {})}}


def api_path_record(circuit: dict, device: int, engine, invalid: list) -> dict:
    """What the reference's own benchmark times (wenbo_engine/bench/end_to_end.py:14-30, docs/v3_comparison.md:19-62): ONE call
    of the drop-in entry points on the circuit, wall time from the call to the synchronised state -- validation, planning
    (cold: nothing cached for this op list) and execution, no repeats, no layout search:
      single_node.run(cd, None, chunk_size = 2^n, use_fusion = True)   (wenbo_engine/runner/single_node.py:78-138)
      Driver().run_circuit(cd)                                          (v3_hisvsim_spark/src/driver.py:135-220)
    plus the amplitudes of both against the engine's identity-layout execution (layout-aware fingerprints)."""
    from quantum_simulations_amd.driver import Driver
    from quantum_simulations_amd.runner import single_node
    n = circuit["number_of_qubits"]
    n_gates = len(circuit["gates"])
    seed = 20260504
    engine.init_zero_state()
    saved_mode, engine.layout_mode = engine.layout_mode, "identity"
    engine.execute(engine.plan(circuit, repeats=1))
    engine.layout_mode = saved_mode
    want = engine.state.fingerprint(n, 0, None, seed)
    rec = {}
    from quantum_simulations_amd import _lib
    _lib.check(_lib.load().qsim_plan_cache_clear())      # COLD: the library plans the op list inside the timed call
    t0 = time.perf_counter()
    buf = single_node.run(circuit, None, chunk_size=1 << n, use_fusion=True, device=device)
    buf.state.sync()
    dt = time.perf_counter() - t0
    diff = abs(buf.state.fingerprint(n, 0, None, seed) - want)
    rec["single_node.run"] = {"seconds": round(dt, 4), "gate_apps_per_s": round(n_gates / dt, 1), "hbm_passes": buf.stats["hbm_passes"],
                              "steps": buf.stats["steps"], "fingerprint_abs_diff_vs_engine": diff,
                              "call": f"single_node.run(cd, None, chunk_size=1<<{n}, use_fusion=True)"}
    buf.close()
    if not diff < PARITY_TOL:
        invalid.append(f"api_path single_node.run: fingerprint differs from the engine's by {diff:.3e}")
    _lib.check(_lib.load().qsim_plan_cache_clear())
    t0 = time.perf_counter()
    with Driver(device=device) as drv:
        res = drv.run_circuit(circuit)
        dt = time.perf_counter() - t0
        diff = abs(res.final_state.fingerprint(n, 0, None, seed) - want)
        rec["Driver.run_circuit"] = {"seconds": round(dt, 4), "elapsed_time_reported": round(res.elapsed_time, 4),
                                     "gate_apps_per_s": round(n_gates / dt, 1), "n_levels": res.n_levels,
                                     "fingerprint_abs_diff_vs_engine": diff, "call": "Driver().run_circuit(cd)"}
        res.final_state.close()
    if not diff < PARITY_TOL:
        invalid.append(f"api_path Driver.run_circuit: fingerprint differs from the engine's by {diff:.3e}")
    rec["note"] = ("one call each, cold, end to end by the host clock (validation + planning + execution + sync); no layout "
                   "search, no repeats: the one-shot cost next to the steady-state `value`")
    return rec


# ---------------------------------------------------------------------------------- N = 1
def run_single(args, k: int) -> tuple[dict, list[str]]:
    from quantum_simulations_amd.circuits import random_1q_cx_circuit
    from quantum_simulations_amd.runner.engine import make_engine

    invalid: list[str] = []
    n = k
    circuit = random_1q_cx_circuit(n, depth=args.depth)
    n_gates = len(circuit["gates"])
    # (the timed steps stand for a run that repeats its plan many times: the layout search is made as for such a run,
    # whatever --steps / --warmup are)
    engine = make_engine(n, 1, 0, int(os.environ.get("LOCAL_RANK", "0")), mode=args.mode,
                         layout="search" if args.layout == "auto" else args.layout, tune_on_device=args.layout == "auto")
    engine.init_zero_state()
    engine.barrier()
    t_plan = time.perf_counter()
    plan = engine.plan(circuit, repeats=args.warmup + args.steps)
    plan_seconds = time.perf_counter() - t_plan
    # the FIRST execution of the plan, by the host clock (cold pass images, first launches): what a caller who runs the
    # circuit once waits for after planning -- `value` below is the steady state of the repeated plan
    first_execution_ms = None
    if args.warmup > 0:                       # (the first of the W warm-up steps; with --warmup 0 the timed region starts cold)
        engine.barrier()
        t_first = time.perf_counter()
        engine.execute(plan)
        engine.barrier()
        first_execution_ms = (time.perf_counter() - t_first) * 1e3
    for _ in range(max(0, args.warmup - 1)):
        engine.execute(plan)
    engine.barrier()
    engine.profile_begin()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        engine.execute(plan)
    engine.barrier()
    dt = time.perf_counter() - t0
    prof = engine.profile_end()
    norm2 = engine.norm2()
    passes = engine.passes_per_step(plan)
    if abs(norm2 - 1.0) > NORM_TOL:
        invalid.append(f"|norm2 - 1| = {abs(norm2 - 1.0):.3e} > {NORM_TOL}")

    # a second, longer timed region (no per-launch events): the same step repeated for ~sustain_seconds
    sustained = None
    if args.sustain_seconds > 0:
        reps = max(args.steps, int(args.sustain_seconds / max(dt / args.steps, 1e-6)))
        engine.barrier()
        t1 = time.perf_counter()
        for _ in range(reps):
            engine.execute(plan)
        engine.barrier()
        dt_s = time.perf_counter() - t1
        sustained = {"steps": reps, "seconds": round(dt_s, 3), "gate_apps_per_s": round(n_gates * reps / dt_s, 1),
                     "ms_per_step": round(dt_s / reps * 1e3, 3)}

    # the TIMED plan itself against an identity-layout execution of the same circuit (the prefix parity below
    # plans its prefix anew; a wrong but unitary timed plan -- a bad tile mask, a bad layout composition -- would pass the
    # norm check): one execution of `plan` from |0..0>, its layout-aware fingerprint, and the same of a plan made with
    # layout = "identity" (index bit = qubit, the library's own tiles)
    timed_plan_check = None
    if args.mode == "fused" and not args.no_plan_check:
        seed = 20260504
        engine.init_zero_state()
        engine.execute(plan)
        fp_timed = engine.state.fingerprint(n, 0, engine.l2p, seed)
        saved_mode, engine.layout_mode = engine.layout_mode, "identity"
        engine.init_zero_state()
        plain = engine.plan(circuit, repeats=1)
        engine.execute(plain)
        fp_plain = engine.state.fingerprint(n, 0, engine.l2p, seed)
        engine.layout_mode = saved_mode
        timed_plan_check = {"fingerprint_abs_diff_vs_identity_layout_plan": abs(fp_timed - fp_plain),
                            "passes_identity_layout_plan": engine.passes_per_step(plain)}
        if not abs(fp_timed - fp_plain) < PARITY_TOL:
            invalid.append(f"the timed plan's state differs from an identity-layout execution of the same circuit: "
                           f"fingerprints {abs(fp_timed - fp_plain):.3e} apart > {PARITY_TOL}")

    copy = engine.copy_ceiling()                     # same-run device-to-device copy of a same-size buffer
    stream = engine.stream_ceiling()                 # ... and the best of everything that streams (copies, in-place RMW)
    dom = max(prof, key=lambda e: e["total_ms"]) if prof else None
    roofline = None
    if dom:
        secs = dom["total_ms"] * 1e-3
        moved = dom["hbm_bytes"] / secs / 1e9
        traffic, traffic_source = pmc_traffic_per_launch(dom["kernel"].split(" ")[0], k)
        roofline = {"bound": "hbm", "achieved": round(moved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(moved / HBM_PEAK_GBS, 4),
                    "traffic": traffic, "traffic_source": traffic_source,
                    "kernel": dom["kernel"], "launches": dom["launches"],
                    "avg_launch_ms": round(dom["total_ms"] / dom["launches"], 4),
                    "bytes_per_launch": dom["hbm_bytes"] / dom["launches"],
                    "copy_ceiling_GBps": copy["GBps"], "frac_of_copy_ceiling": round(moved / copy["GBps"], 4),
                    # against what this box was SEEN to stream in this run (max of copy kernels, hipMemcpy, in-place RMW)
                    "stream_ceiling_GBps": stream["GBps"], "frac_of_achievable": round(moved / stream["GBps"], 4),
                    "frac_x_passes": round(moved / HBM_PEAK_GBS * passes, 3),
                    "gates_per_launch": round(n_gates * args.steps / dom["launches"], 2),
                    "algorithmic_GBps": round(dom["algorithmic_bytes"] / secs / 1e9, 1),
                    "note": "achieved = bytes the launch itself reads+writes (32 B per amplitude of the shard) / "
                            "HIP-event time of the launches in the timed region; algorithmic_GBps sums SURVEY 8d's "
                            "per-gate bytes over the gates a fused launch applies (it may exceed the physical peak "
                            "and is not a roofline fraction)"}
    layout_rec = ("identity" if getattr(plan, "l2p", None) is None else
                  {"logical_to_index_bit": plan.l2p, **getattr(plan, "layout_info", {})})
    out = {
        "metric": "gate-applications/sec (random 1q+CX circuit, complex128 statevector)",
        "value": round(n_gates * args.steps / dt, 2), "unit": "gate-applications/s",
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{n}-qubit random 1q+CX circuit depth {args.depth} (seed 20260228), "
                               f"{n_gates} gates, complex128, {k} local qubits per GPU",
                   "n_qubits": n, "local_qubits": k, "gates_per_step": n_gates, "mode": args.mode,
                   "hbm_passes_per_step": passes,
                   # which index bit a qubit lives on: chosen per plan from a measured model of the tiles' DRAM pattern
                   # (runner/tile_layout.py); the passes are the same, the state is held in that layout
                   "qubit_layout": layout_rec,
                   "value_is": "the steady state of a plan that is executed repeatedly (layout search, planning and the first, "
                               "cold execution are outside the timed region: `plan_seconds`, `first_execution_ms`); what ONE call "
                               "of the drop-in entry points costs end to end is `api_path`"},
        "timed_seconds": round(dt, 4),
        # which index bit a qubit lives on, and what choosing it cost (host search + timing runs on the device), at the top
        # level too: `value` is the steady state AFTER that search
        "layout": "identity" if layout_rec == "identity" else "searched",
        "qubit_layout_seconds": None if layout_rec == "identity" else layout_rec.get("seconds"),
        "plan_seconds": round(plan_seconds, 3), "first_execution_ms": None if first_execution_ms is None else round(first_execution_ms, 3),
        "amplitude_updates_per_s": n_gates * args.steps * float(1 << n) / dt,
        "sustained": sustained,
        "norm2_after": norm2,
        "timed_plan_check": timed_plan_check,
        "roofline": roofline,
        "copy_ceiling": copy,
        "stream_ceiling": stream,
        "kernel_breakdown": [{**e, "total_ms": round(e["total_ms"], 3)} for e in prof],
    }
    if args.fused_qubits and args.fused_qubits != n and args.mode == "fused":
        # the same workload family at the largest single-GPU size of BASELINE.json's list (16 GiB state): ~2 s of GPU
        # time, so the 30-qubit figure of the fused pass is timed by the driver's run and not only by profiles/
        out["fused%d" % args.fused_qubits] = fused_run(args.fused_qubits, args.depth, steps=5, warmup=2,
                                                        device=int(os.environ.get("LOCAL_RANK", "0")), layout=args.layout)
        if abs(out["fused%d" % args.fused_qubits]["norm2_after"] - 1.0) > NORM_TOL:
            invalid.append(f"fused{args.fused_qubits}: norm check failed")
    if not args.no_api_path:
        out["api_path"] = api_path_record(circuit, int(os.environ.get("LOCAL_RANK", "0")), engine, invalid)
    if not args.no_sweep:
        # BASELINE config 3 / north-star target: one gate per launch on a 30-qubit random state (no fusion
        # across the timed gates), every target index, fractions of the 8 TB/s peak of SURVEY 8d's
        # algorithmic bytes (H 32 N, T and CNOT 16 N)
        sw = engine.sweep_gates(args.sweep_qubits)
        out["sweep30"] = sw
        h = sw["rows"]["H(q)"]
        out["roofline_per_gate_kernel"] = {
            "bound": "hbm", "kernel": "k_gate<2> / k_gate_shuffle<1,1>: dense 1q, one launch per gate",
            "achieved": round(h["median_frac"] * HBM_PEAK_GBS, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": h["median_frac"], "min_frac_over_targets": h["min_frac"],
            "workload": f"H on every target of a {sw['n_qubits']}-qubit state (32 B x 2^n per launch)"}
    if not args.no_cpu_baseline:
        base, psi_cpu, done = cpu_baseline(circuit, args.cpu_seconds)
        out["cpu_baseline"] = base
        out["vs_cpu_baseline"] = round(out["value"] / base["value"], 1)
        # parity of the same prefix on the GPU (outside every timed region)
        par = engine.prefix_parity(circuit, done, psi_cpu)
        out["parity_max_abs_diff_vs_cpu_prefix"] = par
        out["parity_path"] = ("prefix planned like the timed steps: layout search + named tiles, compared through the layout"
                              if getattr(engine, "last_parity_layout", None) else "prefix through the fused path, identity layout")
        if not par <= PARITY_TOL:
            invalid.append(f"parity vs CPU prefix {par:.3e} > {PARITY_TOL}")
        del psi_cpu
        out["cpu_baseline_config1"] = cpu_config1_columns()
        for name, col in out["cpu_baseline_config1"].items():
            if not col["amplitudes_exact"]:
                invalid.append(f"config-1 CPU column {name}: wrong GHZ amplitudes")
    engine.close()
    return out, invalid


# ---------------------------------------------------------------------------------- N > 1
def run_multi(args, world: int, rank: int, local_rank: int, k: int) -> tuple[dict | None, list[str]]:
    from quantum_simulations_amd import circuits as gen
    from quantum_simulations_amd.runner.engine import make_engine

    invalid: list[str] = []
    p = world.bit_length() - 1
    n = k + p
    if rank == 0:
        # a line a minute on stderr while the ranks work (the one-device reference runs of 2^33 amplitudes take a while; a
        # silent job looks hung to whoever watches it)
        import threading
        t_start = time.time()

        def heartbeat():
            while True:
                time.sleep(60)
                print(f"bench.py: {time.time() - t_start:.0f} s, {world} ranks at work", file=sys.stderr, flush=True)
        threading.Thread(target=heartbeat, daemon=True).start()
    circuit = gen.random_1q_cx_circuit(n, depth=args.depth)
    n_gates = len(circuit["gates"])
    engine = make_engine(n, world, rank, local_rank, mode=args.mode, rehearsal=args.rehearsal, exchange=args.exchange,
                         pipeline_relayout=not args.no_relayout_pipeline)
    budget = SectionBudget(args.budget_seconds, t0=T0, agree=engine.max_over_ranks,
                           log=(lambda msg: print(msg, file=sys.stderr, flush=True)) if rank == 0 else (lambda msg: None))
    budget.begin("plan (start layout search, stage boundaries + tile passes of every execution)")
    engine.init_zero_state()
    ab_steps = max(1, args.ab_steps)
    t_plan = time.perf_counter()
    plan = engine.plan(circuit, repeats=args.warmup + args.steps + 2 * ab_steps)
    plan_seconds = time.perf_counter() - t_plan
    layout_info = getattr(engine, "layout_info", None)
    budget.begin("warm-up + the timed steps")
    for _ in range(args.warmup):
        engine.execute(plan)
    engine.barrier()
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
    xgmi = engine.comm_stats()
    passes = engine.passes_per_step(plan)
    if abs(norm2 - 1.0) > NORM_TOL:
        invalid.append(f"|norm2 - 1| = {abs(norm2 - 1.0):.3e} > {NORM_TOL}")
    dom0 = max(prof, key=lambda e: e["total_ms"]) if prof else None
    pass_ms = dom0["total_ms"] / dom0["launches"] if dom0 else None

    def timed_steps(count: int) -> float:
        engine.barrier()
        t = time.perf_counter()
        for _ in range(count):
            engine.execute(plan)
        engine.barrier()
        return engine.max_over_ranks(time.perf_counter() - t) / count * 1e3

    # ---- what a re-layout costs on THIS machine, by m (replaces the modelled RELAYOUT_PASSES of the planner) ----------
    relayout_measured = None
    if budget.begin("re-layout measurements (m = 1 .. p, there and back)", optional=True):
        relayout_measured = engine.measure_relayouts(reps=1)
        for rec in relayout_measured:
            ms = rec["exchange_event_ms"] if rec["exchange_event_ms"] is not None else None
            rec["relayout_in_pass_units"] = round(ms / pass_ms, 2) if (ms is not None and pass_ms) else None
            rec["modelled_pass_units"] = engine.RELAYOUT_PASSES.get(rec["m"])
    # ---- re-layouts fused into the neighbouring passes: on (the timed region) / off ------------------------------------
    fused_ab = None
    if budget.begin("fused re-layout on / off", optional=True):
        on_ms = timed_steps(ab_steps)
        engine.fuse_relayout = False
        off_ms = timed_steps(ab_steps)
        off_passes = engine.passes_per_step(plan)
        engine.fuse_relayout = True
        fused_ab = {"steps_each": ab_steps, "ms_per_step_fused": round(on_ms, 3), "ms_per_step_unfused": round(off_ms, 3),
                    "hbm_passes_fused": passes, "hbm_passes_unfused": off_passes}
        n2 = engine.norm2()
        if abs(n2 - 1.0) > NORM_TOL:
            invalid.append(f"after the fused on / off region: |norm2 - 1| = {abs(n2 - 1.0):.3e}")

    configs = None
    if not args.no_configs:
        budget.begin("configs 4 / 5 and the amplitude checks against one-GPU runs")
        # config 5 closed forms; config 4 staged / unstaged and the random 1q+CX circuit: every amplitude, through
        # per-shard fingerprints, against a one-GPU run of the same circuit on rank 0
        configs = engine.run_baseline_configs(gen, check_amplitudes=not args.no_amplitude_check)
        for rec in configs.get("config5", []):
            if not rec["max_abs_err_vs_closed_form"] < PARITY_TOL:
                invalid.append(f"config 5 {rec['circuit']}: max error {rec['max_abs_err_vs_closed_form']:.3e}")
            if not rec.get("max_abs_err_sampled_host_check", 0.0) < PARITY_TOL:
                invalid.append(f"config 5 {rec['circuit']}: sampled host check {rec['max_abs_err_sampled_host_check']:.3e}")
        for key in ("config4", "random_1q_cx"):
            rec = configs.get(key) or {}
            for label in ("staged", "unstaged"):
                run = rec.get(label)
                if not run:
                    continue
                if abs(run["norm2"] - 1.0) > NORM_TOL:
                    invalid.append(f"{key} {label}: norm check failed")
                d = run.get("fingerprint_max_abs_diff_vs_single_gpu")
                if d is not None and not d < PARITY_TOL:
                    invalid.append(f"{key} {label}: shard fingerprints differ from the one-GPU run by {d:.3e} > {PARITY_TOL}")
    # the same workload family on ONE GPU at the same local size (rank 0, outside the timed region, the other
    # ranks wait): per-GPU work is what weak scaling holds fixed, and the N = 1 default of this script is the
    # 28-qubit metric configuration, not a 30-local-qubit one
    single = None
    run_single_ref = (not args.no_single_reference) and budget.begin("one GPU at the same local size (rank 0)", optional=True)
    if rank == 0 and run_single_ref:
        from quantum_simulations_amd.runner.engine import SingleGpuEngine
        e1 = SingleGpuEngine(k, device=local_rank, mode=args.mode)
        c1 = gen.random_1q_cx_circuit(k, depth=args.depth)
        e1.init_zero_state()
        reps1 = max(6, min(args.steps, 10))          # (2 + 6 executions: enough for the engine's layout search, as in the N = 1 line)
        p1 = e1.plan(c1, repeats=2 + reps1)
        for _ in range(2):
            e1.execute(p1)
        e1.barrier()
        t1 = time.perf_counter()
        for _ in range(reps1):
            e1.execute(p1)
        e1.barrier()
        dt1 = time.perf_counter() - t1
        single = {"n_qubits": k, "gates_per_step": len(c1["gates"]), "steps": reps1,
                  "ms_per_step": round(dt1 / reps1 * 1e3, 3), "hbm_passes_per_step": e1.passes_per_step(p1),
                  "gate_apps_per_s": round(len(c1["gates"]) * reps1 / dt1, 2),
                  "amplitude_updates_per_s": len(c1["gates"]) * reps1 * float(1 << k) / dt1}
        e1.close()
    engine.barrier()
    out = None
    if rank == 0:
        out = multi_line(args, world, k, n, n_gates, dt, prof, passes, layout_info, plan_seconds, relayout_measured, fused_ab, single,
                         norm2, xgmi, configs, engine)
    # ---- the OTHER exchange API on the same schedule (torch P2P <-> the library's own communicator) ---------------------
    # It runs LAST, when the line of everything above is complete, under a watchdog: a path that has never run on more than
    # one GPU may hang in a collective, and a hung job must not lose the measurements it already has -- after
    # `--other-api-timeout` seconds rank 0 prints the line as it stands (the section marked as abandoned) and every rank exits.
    import threading
    other_api = "cabi" if args.exchange == "torch" else "torch"
    other = {"exchange_api": other_api, "ran": False}
    section_done, shutdown_done = threading.Event(), threading.Event()

    def abandon(event, seconds: float, what: str):
        """The line must not be lost to a hang: when `event` is not set within `seconds`, rank 0 prints what it has and
        every rank leaves (os._exit: a rank stuck in a collective cannot be joined)."""
        if event.wait(seconds):
            return
        if rank == 0:
            other.setdefault("error", f"no result after {seconds:.0f} s: {what} was abandoned (a hang in a collective?)")
            out["other_exchange_api"] = other
            out["wall_clock"] = budget.report()
            if invalid:
                out["invalid"] = invalid
            print(json.dumps(out), flush=True)
        os._exit(1 if invalid else 0)
    hang_hook = os.environ.get("BENCH_TEST_OTHER_API_HANG") == "1"      # (tests/test_gpu_bench_launcher.py: a section that never returns)
    if args.rehearsal and other_api == "cabi" and not hang_hook:
        other["skipped"] = "rehearsal: the library's RCCL communicator needs one rank per GPU"
    elif budget.begin(f"the other exchange API ({other_api}): GHZ+QFT closed form + {max(1, args.ab_steps)} timed steps", optional=True):
        threading.Thread(target=abandon, args=(section_done, args.other_api_timeout, "the section"), daemon=True).start()
        ab_steps = max(1, args.ab_steps)
        try:
            from quantum_simulations_amd.runner.distributed import DistributedEngine
            engine._flush_local()
            engine.barrier()
            if hang_hook:
                time.sleep(10 ** 6)
            if hasattr(engine.backend, "release_buffers"):
                engine.backend.release_buffers()                     # (room for the second engine's three buffers)
            e2 = DistributedEngine(n, world, rank, local_rank, mode=args.mode, exchange=other_api, init_process_group=False,
                                   pipeline_relayout=not args.no_relayout_pipeline)
            qft = gen.generate_ghz_qft(n)
            e2.init_zero_state()
            e2.execute(e2.plan(qft))
            err = e2.closed_form_error("ghz_qft")
            e2.init_zero_state()
            plan2 = e2.plan(circuit, repeats=1 + ab_steps)
            e2.execute(plan2)
            e2.barrier()
            t2 = time.perf_counter()
            for _ in range(ab_steps):
                e2.execute(plan2)
            e2.barrier()
            ms2 = e2.max_over_ranks(time.perf_counter() - t2) / ab_steps * 1e3
            n2 = e2.norm2()
            other.update(ran=True, ms_per_step=round(ms2, 3), gate_apps_per_s=round(n_gates / (ms2 * 1e-3), 2), steps=ab_steps,
                         ghz_qft_max_abs_err_vs_closed_form=err, norm2=n2, hbm_passes_per_step=e2.passes_per_step(plan2),
                         ok=bool(err < PARITY_TOL and abs(n2 - 1.0) < NORM_TOL))
            e2.backend.close()
        except Exception as e:                                       # noqa: BLE001 -- reported, not fatal
            other["error"] = f"{type(e).__name__}: {e}"[:500]
        section_done.set()
    section_done.set()
    # shutting down is a collective too (a barrier, then the process group goes): if a rank left the section above by an
    # exception the others never saw, it would wait here for ever -- same guard, with the complete line
    if rank == 0:
        out["other_exchange_api"] = other
        out["wall_clock"] = budget.report()
    threading.Thread(target=abandon, args=(shutdown_done, 90.0, "the shutdown"), daemon=True).start()
    engine.close()
    shutdown_done.set()
    return (out if rank == 0 else None), invalid


def multi_line(args, world, k, n, n_gates, dt, prof, passes, layout_info, plan_seconds, relayout_measured, fused_ab, single, norm2, xgmi,
               configs, engine) -> dict:
    """The N > 1 line (rank 0) of everything but the second exchange API."""
    dom = max(prof, key=lambda e: e["total_ms"]) if prof else None
    roofline = None
    kernel_ms = sum(e["total_ms"] for e in prof)
    if dom:
        secs = dom["total_ms"] * 1e-3
        moved = dom["hbm_bytes"] / secs / 1e9
        traffic, traffic_source = pmc_traffic_per_launch(dom["kernel"].split(" ")[0], k)   # a 1-GPU PMC run at this shard size
        roofline = {"bound": "hbm", "achieved": round(moved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(moved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source,
                    "kernel": dom["kernel"], "launches": dom["launches"],
                    "avg_launch_ms": round(dom["total_ms"] / dom["launches"], 4),
                    "bytes_per_launch": dom["hbm_bytes"] / dom["launches"],
                    "algorithmic_GBps": round(dom["algorithmic_bytes"] / secs / 1e9, 1),
                    "note": "rank 0's launches; achieved = bytes a launch reads+writes on its shard / HIP-event time"}
    out = {
        "metric": "gate-applications/sec (random 1q+CX circuit, complex128 statevector)",
        "value": round(n_gates * args.steps / dt, 2), "unit": "gate-applications/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        # what carried the transfers of this line: "rccl" (one rank per GPU over xGMI) or "gloo-rehearsal" (ranks sharing
        # the visible GPUs, host-staged: NOT a multi-GPU number); exchange_api: who posts them (torch P2P / the C ABI)
        "exchange": engine.exchange, "exchange_api": engine.exchange_api, "relayout_pipeline": not args.no_relayout_pipeline,
        "plan_seconds": round(plan_seconds, 3),
        # measured on this machine: an all-to-all over m rank bits (ms, GB/s per rank, in units of one fused pass); the
        # step with the re-layouts NOT fused into the neighbouring passes; the same schedule through the other exchange API
        "relayout_measured": relayout_measured, "fused_relayout_ab": fused_ab, "other_exchange_api": None,
        "exchange_ms_per_step_max_over_ranks": (round(xgmi["exchange_ms_max_over_ranks"] / args.steps, 3)
                                                if xgmi.get("exchange_ms_max_over_ranks") is not None else None),
        "config": {"workload": f"{n}-qubit random 1q+CX circuit depth {args.depth} (seed 20260228), "
                               f"{n_gates} gates, complex128, {k} local qubits per GPU, shards = high qubits",
                   "n_qubits": n, "local_qubits": k, "gates_per_step": n_gates, "mode": args.mode,
                   "hbm_passes_per_step": passes,
                   # the initial qubit layout of the partition (|0..0> is the same state under all of them): the identity
                   # or a random assignment, whichever gives the staged schedule fewer passes + re-layouts (model weights)
                   "qubit_layout": layout_info,
                   "unit_note": "value = gates of the circuit applied to the WHOLE 2^n state per second (the same "
                                "unit at every N; per-GPU shard work is fixed, so this is weak scaling)"},
        "timed_seconds": round(dt, 4),
        "shard_gate_apps_per_s": round(n_gates * world * args.steps / dt, 2),
        # gates x amplitudes of the whole state per second: the size-independent throughput (weak scaling holds
        # the per-GPU share of it fixed); next to it the same on ONE GPU at the same local size, same job
        "amplitude_updates_per_s": n_gates * args.steps * float(1 << n) / dt,
        "single_gpu_same_local_size": single,
        "rank0_gate_kernel_time_share": round(kernel_ms * 1e-3 / dt, 4),
        "norm2_after": norm2,
        "roofline": roofline,
        "xgmi": xgmi,
        "kernel_breakdown": [{**e, "total_ms": round(e["total_ms"], 3)} for e in prof],
        "baseline_configs": configs,
    }
    return out


# ---------------------------------------------------------------------------------- launcher-free N > 1
def launch_ranks(n: int, script=None) -> int:
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment: start the N ranks from here (`script`: another
    entry point started the same way, tools/run_config.py --ranks N).

    The parent (this process) makes NO GPU call -- no torch import, no library load -- picks a free port, starts N fresh
    interpreters of this script (never exec: each child is a new process that initialises its own GPU), relays what rank 0
    prints, and watches all of them: the first rank that exits non-zero ends the run -- the others are terminated (then
    killed) instead of being waited for in a rendezvous or a collective that can no longer complete -- and the parent exits
    with that rank's code.  Nothing is restarted.  Children die with the parent (PR_SET_PDEATHSIG)."""
    import signal
    import socket
    import subprocess
    import threading

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]

    def die_with_parent():
        try:
            import ctypes
            ctypes.CDLL("libc.so.6", use_errno=True).prctl(1, signal.SIGKILL)       # PR_SET_PDEATHSIG
        except Exception:
            pass

    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), BENCH_RANKS_STARTED_BY="bench.py")
        procs.append(subprocess.Popen([sys.executable, str(Path(script or __file__).resolve())] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, preexec_fn=die_with_parent))

    def relay(pipe):
        # rank 0's JSON line(s) go to stdout; whatever else a library prints there (gloo's connection banner) to stderr
        for line in iter(pipe.readline, b""):
            out = sys.stdout.buffer if line.lstrip().startswith(b"{") else sys.stderr.buffer
            out.write(line)
            out.flush()

    relay_thread = threading.Thread(target=relay, args=(procs[0].stdout,), daemon=True)
    relay_thread.start()

    def stop_all():
        for p in procs:
            if p.poll() is None:
                p.terminate()
        deadline = time.time() + 5.0
        for p in procs:
            try:
                p.wait(max(0.1, deadline - time.time()))
            except subprocess.TimeoutExpired:
                p.kill()

    def on_signal(signum, _frame):
        stop_all()
        sys.exit(128 + signum)

    signal.signal(signal.SIGTERM, on_signal)
    signal.signal(signal.SIGINT, on_signal)
    rc = 0
    while True:
        codes = [p.poll() for p in procs]
        bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            r, rc = bad[0]
            print(f"bench.py: rank {r} exited with code {rc}; stopping the other ranks", file=sys.stderr, flush=True)
            stop_all()
            rc = rc if rc > 0 else 1
            break
        if all(c == 0 for c in codes):
            break
        time.sleep(0.1)
    relay_thread.join(5.0)
    return rc


def main():
    args = parse_args()
    refuse_probe_environment()
    if args.gpus < 1 or args.gpus & (args.gpus - 1):
        raise SystemExit("number of GPUs must be a power of two (shards are indexed by high qubits)")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE {world}")
    fail_rank = os.environ.get("BENCH_TEST_FAIL_RANK")          # (tests/test_bench_launcher.py: a rank that dies at start)
    if fail_rank is not None and int(fail_rank) == rank:
        sys.exit(3)
    k = args.local_qubits or (28 if world == 1 else 30)

    if args.dry_run:
        from quantum_simulations_amd.runner.dry_run import main as dry_main
        sys.exit(dry_main(world, rank, k))

    if world == 1:
        out, invalid = run_single(args, k)
    else:
        out, invalid = run_multi(args, world, rank, local_rank, k)
    if out is not None:
        if invalid:
            out["invalid"] = invalid
        print(json.dumps(out), flush=True)
    if invalid:
        sys.exit(1)


if __name__ == "__main__":
    main()
