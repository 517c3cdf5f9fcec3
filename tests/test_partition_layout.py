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


def test_planning_backend_counts_what_the_library_would_do():
    """PlanningBackend.apply_ops = the HBM passes of qsim_apply_ops_io: planned tile passes, + 1 for every end that cannot
    ride in one (slab bits inside a line, a slab bit among the last pass's tile bits, nothing to plan)."""
    from quantum_simulations_amd.runner.distributed import PlanningBackend
    k = 20
    be = PlanningBackend(k)
    H = np.array([[1, 1], [1, -1]]) / np.sqrt(2)
    ops = [([q], H) for q in range(3, 9)]                                  # one fused pass on qubits 3..8
    assert be.apply_ops(ops) == 1
    assert be.apply_ops(ops, src=("buf1", [18, 19])) == 1                  # first pass reads the slabs
    assert be.apply_ops(ops, dst=("buf0", [18, 19], "buf1", 2)) == 1       # last pass stores them
    assert be.apply_ops(ops, src=("buf1", [1, 19])) == 2                   # a slab bit inside a 128-byte line: unpack pass
    assert be.apply_ops(ops, dst=("buf0", [5, 19], "buf1", 0)) == 2        # a slab bit is a tile bit of the last pass: pack pass
    assert be.apply_ops([], src=("buf1", [18]), dst=("buf0", [17], "buf1", 1)) == 2     # nothing to plan: unpack + pack
    assert not be.own_slab_in_state()
    assert be.apply_ops(ops, src=("buf1", [18]), dst=("buf0", [17], "buf1", 1)) == 1 and be.own_slab_in_state()
    many = [([q], H) for q in range(3, 20)] * 2                            # more targets than a tile holds: several passes
    assert be.apply_ops(many, src=("buf1", [18]), dst=("buf0", [17], "buf1", 1)) >= 2 and not be.own_slab_in_state()
    small = PlanningBackend(6)                                             # shards too small for tile passes: a launch per gate
    assert small.apply_ops([([1], H), ([2], H)]) == 2 and small.weight == 2.0


def test_shadow_engine_prices_a_candidate_like_the_real_run():
    """`_candidate_cost` executes the schedule on a PlanningBackend: its pass count equals what a dry engine of the same
    configuration reports for the same circuit and layout (same code path), and the re-layouts are logged by size."""
    n, world = 24, 4
    cd = validate_circuit_dict(gen.random_clifford_t_circuit(n, depth=30))
    eng = _engine(n, world, layout="identity")
    cost, passes, relayouts = eng._candidate_cost(cd, list(range(n)))
    assert passes > 0 and relayouts and all(m in (1, 2) for m in relayouts)
    assert cost == pytest.approx(passes + sum(eng.RELAYOUT_PASSES[m] for m in relayouts))     # (k < 26: every pass weighs 1)
    again = eng._candidate_cost(cd, list(range(n)))
    assert again == (cost, passes, relayouts)                                # the shadow engine starts afresh every time
    eng.staging = False                                                      # swap-and-stay moves are re-layouts too
    assert eng._candidate_cost(cd, list(range(n)))[2]


@pytest.mark.parametrize("n,world,circuit", [(26, 4, "clifft"), (27, 8, "rand"), (26, 4, "ghz_qft")])
def test_choice_is_never_dearer_than_the_identity_and_is_reproducible(n, world, circuit):
    cd = validate_circuit_dict({"clifft": gen.random_clifford_t_circuit(n, depth=30), "rand": gen.random_1q_cx_circuit(n, depth=16),
                                "ghz_qft": gen.generate_ghz_qft(n)}[circuit])
    eng = _engine(n, world, layout="search")
    eng.init_zero_state()
    first = eng.choose_initial_layout(cd, n_candidates=12)
    info = eng.layout_info
    assert sorted(first) == list(range(n))
    assert info["chosen"]["cost_max_over_ranks"] <= info["identity"]["cost_max_over_ranks"] and info["candidates"] == 13
    assert (info["chosen"]["index"] == 0) == (first == list(range(n)))
    assert eng.choose_initial_layout(cd, n_candidates=12) == first          # same inputs, same answer (every rank computes it)
    assert eng._candidate_cost(cd, first)[0] == pytest.approx(info["chosen"]["cost_max_over_ranks"], abs=0.006)


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


def test_auto_needs_real_shard_sizes_and_unstaged_schedules_are_searched_too():
    cd = gen.random_clifford_t_circuit(12, depth=10)
    for kw in ({"layout": "auto"}, {"layout": "identity"}):
        eng = _engine(12, 4, **kw)
        eng.init_zero_state()
        eng.plan(cd)
        assert eng.l2p_planned == list(range(12)) and eng.layout_info is None, kw
    eng = _engine(12, 4, layout="search", staging=False)       # swap-and-stay moves are priced like planned re-layouts
    eng.LAYOUT_CANDIDATES = 6
    eng.init_zero_state()
    eng.plan(cd)
    info = eng.layout_info
    assert info is not None and info["chosen"]["cost_max_over_ranks"] <= info["identity"]["cost_max_over_ranks"]
    assert all(m in (1, 2) for m in info["identity"]["relayouts"])
    with pytest.raises(ValueError):
        _engine(12, 4, layout="best")
