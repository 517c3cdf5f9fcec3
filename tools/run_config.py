#!/usr/bin/env python3
"""BASELINE configs 4 and 5 (multi-GPU parity runs), one rank per GPU:

  python tools/run_config.py --config 4 --ranks 4                   # starts its own ranks (bench.py's launcher)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
      --master-port 29533 tools/run_config.py --config 5            # 33-qubit GHZ, then GHZ+QFT
  ... --nproc-per-node 4 tools/run_config.py --config 4             # 32-qubit Clifford+T depth 60

  --qubits N scales the run down; --rehearsal lets the ranks share one GPU (host-staged gloo exchange: the partition, the
  shard sizes, the fused re-layouts and their pieces are the real ones, only the links are not).  Config 5 checks EVERY
  amplitude against the closed forms of SURVEY 8c on the devices (max-abs-error reduction over all shards, staged layout
  included) at 1e-10; config 4 runs the circuit staged (and, --unstaged, with swap-and-stay moves) and checks the norm and
  -- every amplitude -- the per-shard fingerprints against a ONE-device run of the same circuit on rank 0's GPU
  (qsim_fingerprint; the shards' exchange buffers are released first: 32 qubits = 64 GiB next to 4 x 16 GiB of shards).
Rank 0 prints one JSON line per sub-run (and a progress line per minute on stderr).
"""
import argparse
import json
import os
import sys
import threading
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def timed_run(eng, cd):
    eng.init_zero_state()
    eng.reset_comm_stats()
    plan = eng.plan(cd)
    eng.barrier()
    t0 = time.perf_counter()
    eng.execute(plan)
    eng.barrier()
    dt = eng.max_over_ranks(time.perf_counter() - t0)
    return dt, len(plan.executions[0])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, choices=[4, 5], required=True)
    ap.add_argument("--qubits", type=int, default=0)
    ap.add_argument("--rehearsal", action="store_true")
    ap.add_argument("--ranks", type=int, default=0, help="start this many ranks from here (no launcher)")
    ap.add_argument("--unstaged", action="store_true", help="config 4: also the run without staging (swap-and-stay moves)")
    ap.add_argument("--no-amplitude-check", action="store_true")
    args = ap.parse_args()
    if "WORLD_SIZE" not in os.environ:
        if args.ranks < 1:
            ap.error("start under torch.distributed.run or give --ranks N")
        import bench                                     # (light: the parent makes no GPU call)
        sys.exit(bench.launch_ranks(args.ranks, script=__file__))
    import numpy as np

    from quantum_simulations_amd import circuits as gen
    from quantum_simulations_amd.runner.distributed import DistributedEngine
    world, rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"])
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    n = args.qubits or (33 if args.config == 5 else 32)
    t_start = time.time()
    phase = ["starting"]
    if rank == 0:
        def tick():
            while True:
                time.sleep(60)
                print(f"[run_config] {time.time() - t_start:.0f} s: {phase[0]}", file=sys.stderr, flush=True)
        threading.Thread(target=tick, daemon=True).start()
    eng = DistributedEngine(n, world, rank, local_rank, rehearsal=args.rehearsal)
    out = []
    if args.config == 5:
        for kind, cd in (("ghz", gen.generate_ghz_circuit(n)), ("ghz_qft", gen.generate_ghz_qft(n))):
            phase[0] = kind
            dt, steps = timed_run(eng, cd)
            err = eng.closed_form_error(kind)
            err_host = eng.closed_form_sample_error(kind)
            out.append({"config": 5, "circuit": kind, "n_qubits": n, "n_gpus": world, "gates": len(cd["gates"]), "layout": eng.layout_info,
                        "exchange": eng.exchange, "seconds": round(dt, 4), "gate_apps_per_s": round(len(cd["gates"]) / dt, 1),
                        "steps": steps, "max_abs_err_vs_closed_form": err, "max_abs_err_sampled_host_check": err_host,
                        "pass_1e-10": bool(err < 1e-10 and err_host < 1e-10),
                        "norm2": eng.norm2(), "xgmi": eng.comm_stats()})
    else:
        cd = gen.random_clifford_t_circuit(n, depth=60)
        seed = 20260504
        rec = {"config": 4, "n_qubits": n, "n_gpus": world, "gates": len(cd["gates"]), "local_qubits": eng.k,
               "exchange": eng.exchange, "shard_buffers_per_rank": 3, "bytes_per_buffer": 16 << eng.k}
        runs, labels = [], []
        sample_staged = None
        for label, staging in (("staged", True),) + ((("unstaged", False),) if args.unstaged else ()):
            phase[0] = f"config 4 {label}: executing"
            eng.staging = staging
            dt, steps = timed_run(eng, cd)
            rec[label] = {"seconds": round(dt, 4), "gate_apps_per_s": round(len(cd["gates"]) / dt, 1), "steps": steps, "layout": eng.layout_info,
                          "hbm_passes": eng.last_passes, "home_moves_max_over_ranks": int(eng.max_over_ranks(float(eng.home_moves))), "norm2": eng.norm2(), "xgmi": eng.comm_stats()}
            runs.append((eng.fingerprints(seed), eng.shard_selectors()))
            labels.append(label)
            if n <= 26 and label == "staged":       # small rehearsal sizes: compare the whole state with an unstaged run
                sample_staged = eng.state_vector()
            elif sample_staged is not None:
                rec["max_abs_diff_staged_vs_unstaged"] = float(np.max(np.abs(sample_staged - eng.state_vector())))
        if not args.no_amplitude_check:
            phase[0] = "config 4: one-device run for the fingerprints"
            t0 = time.perf_counter()
            diffs = eng.check_against_single_device(cd, runs, seed)
            rec["single_device_check_seconds"] = round(time.perf_counter() - t0, 2)
            for label, d in zip(labels, diffs):
                rec[label]["fingerprint_max_abs_diff_vs_single_gpu"] = d
                rec[label]["pass_1e-10"] = bool(d < 1e-10)
        out.append(rec)
    bad = any(r.get("pass_1e-10") is False or any(isinstance(v, dict) and v.get("pass_1e-10") is False for v in r.values()) for r in out)
    if rank == 0:
        for rec in out:
            print(json.dumps(rec), flush=True)
    eng.close()
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
