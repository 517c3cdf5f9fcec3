"""bench.py's bookkeeping that does not need a GPU: the PMC traffic figure is taken only from a summary made with the
kernel sources of the running build (a stale figure must not look live), at the right shard size."""
import importlib
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def _bench():
    sys.path.insert(0, str(ROOT))
    return importlib.import_module("bench")


def test_traffic_only_from_a_summary_of_the_same_sources(tmp_path, monkeypatch):
    bench = _bench()
    from quantum_simulations_amd._lib import source_hash
    prof = tmp_path / "profiles"
    prof.mkdir()
    monkeypatch.setattr(bench, "ROOT", tmp_path)
    row = {"kernel": "k_tile", "hbm_bytes_per_launch": 8.59e9}
    (prof / "r01_pmc_summary.json").write_text(json.dumps({"csrc_sha16": "0123456789abcdef", "kernels": [row]}))
    assert bench.pmc_traffic_per_launch("k_tile") == (None, None)                       # stale: other sources
    (prof / "r02_pmc_summary.json").write_text(json.dumps({"kernels": [row]}))           # no hash at all (round 2 files)
    assert bench.pmc_traffic_per_launch("k_tile") == (None, None)
    (prof / "r03_pmc_summary.json").write_text(json.dumps({"csrc_sha16": source_hash(), "kernels": [row]}))
    assert bench.pmc_traffic_per_launch("k_tile") == (8.59e9, "profiles/r03_pmc_summary.json")
    assert bench.pmc_traffic_per_launch("k_gate") == (None, None)                       # another kernel
    assert bench.pmc_traffic_per_launch("k_tile", 30) == (None, None)                   # another shard size
    (prof / "r03b_pmc_summary.json").write_text(json.dumps({"csrc_sha16": source_hash(), "local_qubits": 30,
                                                             "kernels": [dict(row, hbm_bytes_per_launch=3.436e10)]}))
    assert bench.pmc_traffic_per_launch("k_tile", 30) == (3.436e10, "profiles/r03b_pmc_summary.json")
    assert bench.pmc_traffic_per_launch("k_tile", 28)[0] == 8.59e9
    (prof / "broken_pmc_summary.json").write_text("{not json")
    assert bench.pmc_traffic_per_launch("k_tile", 28)[0] == 8.59e9


def test_the_committed_summaries_match_the_committed_sources():
    """The judged profiles of the round were taken with the sources in the tree: bench.py will print their traffic."""
    from quantum_simulations_amd._lib import source_hash
    now = source_hash()
    docs = [json.loads(p.read_text()) for p in (ROOT / "profiles").glob("*_pmc_summary.json")]
    live = [d for d in docs if d.get("csrc_sha16") == now]
    if not {d.get("local_qubits", 28) for d in live} >= {28, 30}:
        import pytest
        pytest.skip("profiles/*_pmc_summary.json are older than csrc/: bench.py prints traffic = null until "
                    "tools/profile_round.sh has run on the GPU with these sources")


def test_probe_environment_is_refused(monkeypatch):
    bench = _bench()
    import pytest
    monkeypatch.setenv("QSIM_PLAN_LOOKAHEAD", "0")
    with pytest.raises(SystemExit) as e:
        bench.refuse_probe_environment()
    assert e.value.code == 2
    monkeypatch.delenv("QSIM_PLAN_LOOKAHEAD")
    monkeypatch.setenv("QSIM_DIST_BACKEND", "gloo")     # round 3's rehearsal switch: now an explicit flag (--rehearsal),
    with pytest.raises(SystemExit):                     # no environment variable changes what a bench line means
        bench.refuse_probe_environment()
    monkeypatch.delenv("QSIM_DIST_BACKEND")
    bench.refuse_probe_environment()
