"""Dry run of the multi-GPU communication schedule (bench.py --gpus N --dry-run, CPU tests).

Every rank runs the REAL `DistributedEngine` -- planning, staging, swap-and-stay moves, pipelined
re-layout pieces -- at the full problem size (30 local qubits, n = 32 / 33) over a `PlanningBackend` that
holds no shard memory and does no arithmetic; each transfer the engine would post is recorded instead
(peer, bytes sent, bytes expected, kind).  The ranks then exchange their records over gloo and rank 0
checks what a first run on real RCCL / xGMI would otherwise have to discover:

  * symmetry: the i-th transfer rank a posts towards rank b has the size rank b expects from a in ITS
    i-th transfer with a (a mismatch is a hang or a truncated receive on RCCL);
  * every send / receive slice lies inside its buffer (checked when the slice is taken);
  * every rank posts the same number of transfer groups in the same order of kinds;
  * bytes per rank match the closed form of each move: a re-layout of m qubits ships (1 - 2^-m) of a shard.
Rank 0 prints one JSON line per workload; the return value is the process exit code.
"""
from __future__ import annotations

import json

from quantum_simulations_amd import circuits as gen
from quantum_simulations_amd.runner.distributed import DistributedEngine, PlanningBackend


def workloads(n: int) -> list:
    return [("bench: random 1q+CX depth 40, 2 executions", gen.random_1q_cx_circuit(n, depth=40), True, 2),
            ("config 4: Clifford+T depth 60, staged", gen.random_clifford_t_circuit(n, depth=60), True, 1),
            ("config 4: Clifford+T depth 60, unstaged (swap-and-stay)", gen.random_clifford_t_circuit(n, depth=60), False, 1),
            ("config 5: GHZ", gen.generate_ghz_circuit(n), True, 1),
            ("config 5: GHZ+QFT", gen.generate_ghz_qft(n), True, 1)]


def check(traces: list, world: int, k: int) -> list[str]:
    """traces[r] = [(kind, peer, sent_bytes, recv_bytes), ...] of rank r -> list of problems."""
    problems = []
    for a in range(world):
        for b in range(a + 1, world):
            ab = [(kind, s, r) for kind, peer, s, r in traces[a] if peer == b]
            ba = [(kind, s, r) for kind, peer, s, r in traces[b] if peer == a]
            if len(ab) != len(ba):
                problems.append(f"ranks {a} and {b} post {len(ab)} vs {len(ba)} transfers with each other")
                continue
            for i, ((ka, sa, ra), (kb, sb, rb)) in enumerate(zip(ab, ba)):
                if sa != rb or sb != ra or ka != kb:
                    problems.append(f"transfer {i} between ranks {a} and {b}: {ka} sends {sa} / expects {ra}, "
                                    f"{kb} sends {sb} / expects {rb}")
    shard = 16 << k
    for r, tr in enumerate(traces):
        if any(s <= 0 or s > shard for _, _, s, _ in tr):
            problems.append(f"rank {r}: a transfer is empty or larger than a shard")
    if len({len(tr) for tr in traces}) != 1:
        problems.append(f"ranks post different numbers of transfers: {[len(tr) for tr in traces]}")
    return problems


def run_world(world: int, rank: int, k: int, emit=print) -> int:
    import torch.distributed as dist
    p = world.bit_length() - 1
    n = k + p
    failed = 0
    eng = None
    for title, cd, staging, repeats in workloads(n):
        # (a PlanningBackend: the dry run also knows the HBM passes every rank's library would make -- qsim_plan_ops on the
        # exact op lists, host only -- so the line carries the compute side of the schedule next to its transfers)
        eng = DistributedEngine(n, world, rank, backend=PlanningBackend(k), staging=staging)
        eng.init_zero_state()
        # (the bench plans its circuit for warmup + steps executions: the full planning effort; two of them are run here)
        plan = eng.plan(cd, repeats=repeats, effort="high" if repeats > 1 else None)
        passes = []
        for _ in range(repeats):
            eng.execute(plan)
            passes.append(eng.last_passes)
        traces = [None] * world
        dist.all_gather_object(traces, eng.trace)
        stats = [None] * world
        dist.all_gather_object(stats, {"bytes": eng.xgmi_bytes_sent, "exchanges": eng.exchanges,
                                       "local_batches": eng.backend.local_passes, "layout": eng.l2p,
                                       "groups_posted": eng.trace_posts, "hbm_passes": passes,
                                       "layout_search": eng.layout_info})
        if rank == 0:
            problems = check(traces, world, k)
            failed += bool(problems)
            kinds = {}
            for kind, _, s, _ in traces[0]:
                kinds.setdefault(kind, [0, 0])
                kinds[kind][0] += 1
                kinds[kind][1] += s
            emit(json.dumps({
                "dry_run": title, "n_qubits": n, "n_gpus": world, "local_qubits": k, "gates": len(cd["gates"]),
                "executions": repeats, "ok": not problems, "problems": problems[:8],
                "per_rank": [{"rank": r, "transfers": len(traces[r]), "bytes_sent": stats[r]["bytes"],
                              "exchanges": stats[r]["exchanges"], "local_batches": stats[r]["local_batches"],
                              # HBM passes of the shard per execution, as the library's host planner plans this rank's op lists
                              "hbm_passes_per_execution": stats[r]["hbm_passes"],
                              # posting points: a fused re-layout posts one group (all peers at once) per piece
                              "groups_posted": stats[r]["groups_posted"]}
                             for r in range(world)],
                "rank0_by_kind": {kd: {"transfers": c, "bytes": b} for kd, (c, b) in kinds.items()},
                "rank0_schedule": [{"kind": kind, "peer": peer, "bytes": s} for kind, peer, s, _ in traces[0][:64]],
                "layout_search_rank0": stats[0]["layout_search"],
                "final_layout_rank0": stats[0]["layout"]}))
    flag = [failed]
    dist.broadcast_object_list(flag, src=0)
    if eng is not None:
        eng.close()
    return 1 if flag[0] else 0


def main(world: int, rank: int, k: int) -> int:
    """bench.py --gpus N --dry-run: rank 0 prints ONE JSON line (all workloads inside)."""
    if world < 2:
        print(json.dumps({"error": "--dry-run needs --gpus N > 1 (gloo, CPU only)"}))
        return 2
    docs = []
    rc = run_world(world, rank, k, emit=lambda line: docs.append(json.loads(line)))
    if rank == 0:
        # A PROJECTION, not a measurement: what the schedules above would take with the pass time measured on one MI355X for
        # exactly these op lists (tools/shard_compute_probe.py: 6.65 ms per pass at 30 local qubits, profiles/r05b_*) and the
        # link model of the planner (2^-m of a shard per peer over its own xGMI link at 0.8 x 153 GB/s, nothing hidden
        # behind compute: an upper bound of the exchange's share).  The first run on a node replaces it.
        PASS_MS_30 = 6.65
        for d in docs:
            shard = 16 << k
            steps = []
            for e in range(d["executions"]):
                passes = max(r["hbm_passes_per_execution"][e] for r in d["per_rank"])
                steps.append(passes * PASS_MS_30 * 2.0 ** (k - 30))
            moved = d["per_rank"][0]["bytes_sent"] / max(1, d["executions"])
            ex = d["per_rank"][0]["exchanges"] / max(1, d["executions"])
            p_bits = world.bit_length() - 1
            link_ms = (moved / max(1e-9, ex)) / ((1 << p_bits) - 1) / (0.8 * 153e9) * 1e3 if ex else 0.0   # (full-width re-layouts: all peers at once)
            step_ms = sum(steps) / len(steps) + ex * link_ms
            d["projection_model_not_measured"] = {"pass_ms_at_30_local_qubits_measured_on_one_gpu": PASS_MS_30, "compute_ms_per_execution": round(sum(steps) / len(steps), 1),
                                                  "exchange_ms_per_execution_upper_bound": round(ex * link_ms, 1), "step_ms": round(step_ms, 1),
                                                  "gate_apps_per_s": round(d["gates"] / (step_ms * 1e-3), 1) if step_ms else None}
        print(json.dumps({"dry_run": True, "n_gpus": world, "local_qubits": k, "n_qubits": k + world.bit_length() - 1,
                          "ok": rc == 0 and all(d["ok"] for d in docs), "exchange": "none (schedule only, gloo control plane)",
                          # the fields a run on devices fills in (bench.py run_multi): measured re-layouts by m, the step with
                          # unfused re-layouts, the same schedule through the other exchange API, the wall-clock sections
                          "relayout_measured": None, "fused_relayout_ab": None, "other_exchange_api": None, "wall_clock": None,
                          "workloads": docs}), flush=True)
    return rc
