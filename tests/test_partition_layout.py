"""The partition's initial qubit layout (DistributedEngine.choose_initial_layout): host-only logic -- the
schedule's price under the identity and under random assignments, the choice, and when a plan may make it."""
import numpy as np
import pytest

from quantum_simulations_amd import circuits as gen
from quantum_simulations_amd.circuit.io import validate_circuit_dict
from quantum_simulations_amd.runner.distributed import DistributedEngine, DryBackend


def _engine(n, world, **kw):
    p = world.bit_length() - 1
    return DistributedEngine(n, world, 0, backend=DryBackend(n - p), init_process_group=False, **kw)


def test_cost_counts_passes_relayouts_and_unfusable_ones():
    eng = _engine(24, 4, layout="identity")
    k = eng.k
    H = np.array([[1, 1], [1, -1]]) / np.sqrt(2)
    SWAP = np.eye(4)[[0, 2, 1, 3]]
    local = [([q], H) for q in range(3, 9)]                     # one fused pass
    steps = [{"local_ops": local, "nonlocal_ops": [([5, k], SWAP), ([6, k + 1], SWAP)]},          # one re-layout, m = 2
             {"local_ops": local, "nonlocal_ops": [([1, k], SWAP)]},                               # m = 1, a line bit: + 2 passes
             {"local_ops": [], "nonlocal_ops": [([7, k], SWAP), ([7, k + 1], SWAP)]}]             # not disjoint: two of m = 1
    cost, passes, groups = eng._schedule_cost(steps)
    assert groups == [2, 1, 1, 1]
    assert passes == 1 + 1 + 2
    assert cost == pytest.approx(passes + eng.RELAYOUT_PASSES[2] + 3 * eng.RELAYOUT_PASSES[1])
    eng.fuse_relayout = False                                    # (unfused engines pay pack + unpack everywhere: no extra term)
    assert eng._schedule_cost(steps)[1] == 2


@pytest.mark.parametrize("n,world,circuit", [(26, 4, "clifft"), (27, 8, "rand"), (26, 4, "ghz_qft")])
def test_choice_is_never_dearer_than_the_identity_and_is_reproducible(n, world, circuit):
    cd = validate_circuit_dict({"clifft": gen.random_clifford_t_circuit(n, depth=30), "rand": gen.random_1q_cx_circuit(n, depth=16),
                                "ghz_qft": gen.generate_ghz_qft(n)}[circuit])
    eng = _engine(n, world, layout="search")
    eng.init_zero_state()
    first = eng.choose_initial_layout(cd, n_candidates=12)
    info = eng.layout_info
    assert sorted(first) == list(range(n))
    assert info["chosen"]["cost"] <= info["identity"]["cost"] and info["candidates"] == 13
    assert (info["chosen"]["index"] == 0) == (first == list(range(n)))
    assert eng.choose_initial_layout(cd, n_candidates=12) == first          # same inputs, same answer (every rank computes it)
    steps, _ = eng._steps_from(cd, first)
    assert eng._schedule_cost(steps)[0] == pytest.approx(info["chosen"]["cost"], abs=0.006)


def test_only_the_first_plan_of_a_fresh_state_chooses():
    n, world = 24, 4
    cd = gen.random_clifford_t_circuit(n, depth=20)
    eng = _engine(n, world, layout="search")
    eng.LAYOUT_CANDIDATES = 8
    assert not eng._fresh
    plan0 = eng.plan(cd)                                          # no initialised state yet: the identity
    assert plan0.start_mappings[0] == list(range(n)) and eng.layout_info is None
    eng.init_zero_state()
    plan1 = eng.plan(cd, repeats=2)
    chosen = list(eng.l2p_planned)
    assert plan1.start_mappings[0] == chosen and eng.layout_info is not None
    plan2 = eng.plan(gen.generate_ghz_circuit(n))                 # a second plan before the first runs keeps the layout:
    assert plan2.start_mappings[0] == chosen                      # both stay executable
    eng.execute(plan1)
    assert eng.l2p_planned == plan1.mappings[0]
    eng.init_zero_state()
    assert eng._fresh and eng.l2p_planned == list(range(n)) and eng.layout_info is None
    eng.relayout([[5, eng.k]])                                    # the state is touched outside a plan: no choice any more
    assert not eng._fresh
    assert eng.plan(cd).start_mappings[0] == list(range(n))


def test_auto_needs_staging_and_real_shard_sizes():
    cd = gen.random_clifford_t_circuit(12, depth=10)
    for kw in ({"layout": "auto"}, {"layout": "identity"}, {"layout": "search", "staging": False}):
        eng = _engine(12, 4, **kw)
        eng.init_zero_state()
        eng.plan(cd)
        assert eng.l2p_planned == list(range(12)) and eng.layout_info is None, kw
    with pytest.raises(ValueError):
        _engine(12, 4, layout="best")
