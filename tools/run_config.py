#!/usr/bin/env python3
"""BASELINE configs 4 and 5 (multi-GPU parity runs), one rank per GPU:

  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
      --master-port 29533 tools/run_config.py --config 5            # 33-qubit GHZ, then GHZ+QFT
  ... --nproc-per-node 4 tools/run_config.py --config 4             # 32-qubit Clifford+T depth 60

  --qubits N scales the run down; --rehearsal lets the ranks share one GPU (host-staged gloo exchange).  Config 5 checks EVERY amplitude against the closed forms of SURVEY 8c on the devices
  (max-abs-error reduction over all shards, staged layout included) at 1e-10; config 4 checks
  the norm and compares staged vs unstaged execution on a sampled set of amplitudes.
Rank 0 prints one JSON line per sub-run.
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from quantum_simulations_amd import circuits as gen  # noqa: E402
from quantum_simulations_amd.runner.distributed import DistributedEngine  # noqa: E402


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
    args = ap.parse_args()
    world, rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"])
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    n = args.qubits or (33 if args.config == 5 else 32)
    eng = DistributedEngine(n, world, rank, local_rank, rehearsal=args.rehearsal)
    out = []
    if args.config == 5:
        for kind, cd in (("ghz", gen.generate_ghz_circuit(n)), ("ghz_qft", gen.generate_ghz_qft(n))):
            dt, steps = timed_run(eng, cd)
            err = eng.closed_form_error(kind)
            out.append({"config": 5, "circuit": kind, "n_qubits": n, "n_gpus": world, "gates": len(cd["gates"]),
                        "seconds": round(dt, 4), "gate_apps_per_s": round(len(cd["gates"]) / dt, 1),
                        "steps": steps, "max_abs_err_vs_closed_form": err, "pass_1e-10": bool(err < 1e-10),
                        "norm2": eng.norm2(), "xgmi": eng.comm_stats()})
    else:
        cd = gen.random_clifford_t_circuit(n, depth=60)
        dt, steps = timed_run(eng, cd)
        norm2, stats = eng.norm2(), eng.comm_stats()
        k = eng.k
        sample_staged = None
        if n <= 26:   # small rehearsal sizes: compare the whole state with an unstaged run
            sample_staged = eng.state_vector()
        eng.staging = False
        dt2, steps2 = timed_run(eng, cd)
        rec = {"config": 4, "n_qubits": n, "n_gpus": world, "gates": len(cd["gates"]), "local_qubits": k,
               "staged": {"seconds": round(dt, 4), "gate_apps_per_s": round(len(cd["gates"]) / dt, 1),
                          "steps": steps, "xgmi": stats, "norm2": norm2},
               "unstaged": {"seconds": round(dt2, 4), "gate_apps_per_s": round(len(cd["gates"]) / dt2, 1),
                            "steps": steps2, "xgmi": eng.comm_stats(), "norm2": eng.norm2()}}
        if sample_staged is not None:
            rec["max_abs_diff_staged_vs_unstaged"] = float(np.max(np.abs(sample_staged - eng.state_vector())))
        out.append(rec)
    if rank == 0:
        for rec in out:
            print(json.dumps(rec), flush=True)
    eng.close()


if __name__ == "__main__":
    main()
