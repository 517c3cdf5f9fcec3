"""The multi-GPU communication schedule at FULL problem size without shard memory (`bench.py --gpus N
--dry-run`): world 4 (n = 32, config 4) and world 8 (n = 33, config 5) over gloo on the CPU.  Every rank runs
the real DistributedEngine over DryBackend; rank 0 checks send / receive symmetry between every pair of
ranks, slice bounds and byte counts (quantum_simulations_amd/runner/dry_run.py)."""
import json
import os
import sys
import traceback
from pathlib import Path

import pytest
import torch.multiprocessing as mp

from tests.test_distributed_gloo import _free_port

ROOT = Path(__file__).resolve().parent.parent


def _worker(rank, world, port, k, lines, errors):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        sys.path.insert(0, str(ROOT))
        from quantum_simulations_amd.runner import dry_run
        rc = dry_run.run_world(world, rank, k, emit=lines.put)
        assert rc == 0, "the dry run reported a schedule problem"
    except Exception:
        errors.put((rank, traceback.format_exc()))
        raise


@pytest.mark.parametrize("world", [4, 8])
def test_schedule_is_symmetric_at_full_size(world):
    ctx = mp.get_context("spawn")
    lines, errors = ctx.SimpleQueue(), ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, 30, lines, errors)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
    msgs = []
    while not errors.empty():
        msgs.append(errors.get())
    assert not msgs and all(p.exitcode == 0 for p in procs), "\n".join(f"[rank {r}] {m}" for r, m in msgs)
    docs = []
    while not lines.empty():
        docs.append(json.loads(lines.get()))
    assert len(docs) == 5 and all(d["ok"] and d["n_qubits"] == 30 + world.bit_length() - 1 for d in docs)
    shard = 16 << 30
    by_title = {d["dry_run"].split(":")[0] + ":" + d["dry_run"].split(":")[1][:14]: d for d in docs}
    for d in docs:
        for kind, agg in d["rank0_by_kind"].items():
            assert kind in ("relayout", "swap-and-stay")
            assert 0 < agg["bytes"] <= agg["transfers"] * shard
        for r in d["per_rank"]:
            # a re-layout is posted piece by piece -- one group of all its peers per piece, the next piece being computed
            # meanwhile: at 30 local qubits every re-layout has at least two posting points
            assert r["groups_posted"] >= 2 * r["exchanges"] > 0 or r["exchanges"] == 0, r
        sent = {r["bytes_sent"] for r in d["per_rank"]}
        assert len(sent) == 1, "every rank ships the same number of bytes"        # (the schedule is symmetric)
        # the compute side: every rank's HBM passes per execution as the library's host planner plans ITS op lists (the ranks
        # differ by their rank-bit phases and conditional gates: a pass or two), and the layout search that preceded it
        for r in d["per_rank"]:
            assert len(r["hbm_passes_per_execution"]) == d["executions"] and all(x > 0 for x in r["hbm_passes_per_execution"])
        spread = [max(r["hbm_passes_per_execution"][e] for r in d["per_rank"]) - min(r["hbm_passes_per_execution"][e] for r in d["per_rank"])
                  for e in range(d["executions"])]
        assert max(spread) <= 4, spread
        info = d["layout_search_rank0"]
        assert info is not None and info["chosen"]["cost_max_over_ranks"] <= info["identity"]["cost_max_over_ranks"]
        assert "search_seconds" in info                      # (the host time of the search is in the record)
        # what was priced is what ran: the start layouts are priced with one thin-pass threshold, the plan that runs is the
        # best over several (staging method "tiles"), so it may need a pass less, never more than one more
        ran = d["per_rank"][0]["hbm_passes_per_execution"][0]
        assert ran <= info["chosen"]["passes_this_rank"] + 1, (ran, info["chosen"])
        if d["dry_run"].startswith("bench") or "staged" in d["dry_run"].split(",")[-1] and "unstaged" not in d["dry_run"]:
            # stage boundaries and tile passes planned together: EVERY rank runs the planned passes (the planner names its
            # tiles to the library), so the ranks agree exactly
            assert max(spread) == 0, spread
    # passes per execution of a 30-local-qubit shard (r04: 33 / 32 at 33 qubits on 8 ranks, 28 at 32 on
    # 4; Clifford+T staged 21) with no more re-layouts and no more bytes than before
    bench = next(d for d in docs if d["dry_run"].startswith("bench"))
    cliff = next(d for d in docs if d["dry_run"] == "config 4: Clifford+T depth 60, staged")
    worst = max(max(r["hbm_passes_per_execution"]) for r in bench["per_rank"])
    assert worst <= (26 if world == 8 else 23), worst
    assert bench["per_rank"][0]["exchanges"] <= 8 and bench["per_rank"][0]["bytes_sent"] <= 8 * (shard - (shard >> (world.bit_length() - 1)))
    assert max(max(r["hbm_passes_per_execution"]) for r in cliff["per_rank"]) <= 18 and cliff["per_rank"][0]["exchanges"] <= 3
    # GHZ needs exactly one re-layout of all global qubits: (1 - 2^-p) of a shard per rank
    ghz = next(d for d in docs if d["dry_run"] == "config 5: GHZ")
    p = world.bit_length() - 1
    assert ghz["per_rank"][0]["bytes_sent"] == shard - (shard >> p) and ghz["per_rank"][0]["exchanges"] == 1
    del by_title


def test_dry_check_catches_asymmetry():
    from quantum_simulations_amd.runner.dry_run import check
    good = [[("relayout", 1, 64, 64)], [("relayout", 0, 64, 64)]]
    assert check(good, 2, 3) == []
    bad = [[("relayout", 1, 64, 64)], [("relayout", 0, 32, 64)]]
    assert check(bad, 2, 3)
    missing = [[("relayout", 1, 64, 64)], []]
    assert check(missing, 2, 3)
