"""SingleGpuEngine with a chosen qubit layout (runner/engine.py, runner/tile_layout.py) against the oracle: circuit
families at sizes the oracle finishes in seconds (the layout search forced on: it is for >= 26 qubits by default; the
2^28-amplitude test in test_gpu_kernels.py covers the default path), repeated execution in the layout, a second plan with
another layout on the used state (the SWAP path), fingerprints and logical indices through the layout."""
import numpy as np
import pytest

from oracle import dense_oracle as orc
from quantum_simulations_amd import circuits as gen
from quantum_simulations_amd.circuit.io import validate_circuit_dict
from tests.cpu_shard_backend import fingerprint_np

pytestmark = pytest.mark.gpu


def _engine(n):
    from quantum_simulations_amd.runner.engine import SingleGpuEngine
    eng = SingleGpuEngine(n, layout="search")
    eng.LAYOUT_MIN_QUBITS = 8
    eng.LAYOUT_CANDIDATES = 24
    return eng


@pytest.mark.parametrize("n", [14, 17, 20])
def test_circuit_families_in_a_chosen_layout(n):
    eng = _engine(n)
    families = {"random": gen.random_1q_cx_circuit(n, depth=16, seed=n), "clifford_t": gen.random_clifford_t_circuit(n, depth=30, seed=n),
                "ghz_qft": gen.generate_ghz_qft(n), "qpe": gen.generate_qpe_circuit(n - 1), "w_qft": gen.generate_w_qft(n)}
    layouts = set()
    for name, cd in families.items():
        if cd["number_of_qubits"] != n:
            continue
        want = orc.simulate(validate_circuit_dict(cd))
        eng.init_zero_state()
        plan = eng.plan(cd)
        assert plan.layout_info["passes_chosen"] <= plan.layout_info["passes_identity"]
        eng.execute(plan)
        assert eng.last_passes == plan.layout_info["passes_chosen"], name
        layouts.add(tuple(eng.l2p or range(n)))
        np.testing.assert_allclose(eng.state_vector(), want, rtol=0, atol=1e-10, err_msg=name)
        # physical range -> logical indices, and the layout-aware fingerprint
        got = eng.state.download(1 << (n - 2), 1 << 8)
        np.testing.assert_allclose(got, want[eng.logical_index(1 << (n - 2), 1 << 8)], rtol=0, atol=1e-10)
        assert abs(eng.state.fingerprint(n, 0, eng.l2p, 3) - fingerprint_np(want, n, 0, None, 3)) < 1e-10
        # the same plan again, in place: psi2 = C(C|0>)
        eng.execute(plan)
        want2 = want.copy()
        for g in validate_circuit_dict(cd)["gates"]:
            U = orc.gate_matrix(g["gate"], g["params"])
            (orc.apply_1q if len(g["qubits"]) == 1 else orc.apply_2q)(want2, *g["qubits"], U)
        np.testing.assert_allclose(eng.state_vector(), want2, rtol=0, atol=1e-10, err_msg=name + " (repeat)")
        # another plan, written for another layout, on this used state: the engine moves the qubits first
        other = gen.random_1q_cx_circuit(n, depth=6, seed=1000 + n)
        plan2 = eng.plan(other)
        eng.execute(plan2)
        want3 = want2.copy()
        for g in validate_circuit_dict(other)["gates"]:
            U = orc.gate_matrix(g["gate"], g["params"])
            (orc.apply_1q if len(g["qubits"]) == 1 else orc.apply_2q)(want3, *g["qubits"], U)
        np.testing.assert_allclose(eng.state_vector(), want3, rtol=0, atol=1e-10, err_msg=name + " (second plan)")
        assert abs(eng.norm2() - 1.0) < 1e-10
    assert len(layouts) >= 2                      # (the families really got layouts of their own)
    eng.close()


def test_identity_layout_and_short_plans_skip_the_search():
    from quantum_simulations_amd.runner.engine import SingleGpuEngine
    n = 12
    cd = gen.random_1q_cx_circuit(n, depth=8, seed=3)
    for kw in ({"layout": "identity"}, {"layout": "auto"}):           # auto: 12 qubits are below the threshold
        eng = SingleGpuEngine(n, **kw)
        eng.init_zero_state()
        plan = eng.plan(cd, repeats=100)
        assert plan.l2p is None and all(t is None for t in plan.tiles)
        eng.execute(plan)
        np.testing.assert_allclose(eng.state_vector(), orc.simulate(validate_circuit_dict(cd)), rtol=0, atol=1e-10)
        eng.close()
    with pytest.raises(ValueError):
        SingleGpuEngine(n, layout="best")


def test_tuning_on_the_device_picks_among_minimum_pass_layouts_and_leaves_zero_state():
    """tune_on_device: the minimum-pass layouts are timed on the device -- allowed only while the state is |0..0>, which it
    is again afterwards; on a used state the model's first choice is taken without touching the state."""
    from quantum_simulations_amd.runner.engine import SingleGpuEngine
    n = 18
    eng = SingleGpuEngine(n, layout="search", tune_on_device=True)
    eng.LAYOUT_MIN_QUBITS, eng.LAYOUT_CANDIDATES = 8, 48
    cd = gen.random_1q_cx_circuit(n, depth=20, seed=5)
    want = orc.simulate(validate_circuit_dict(cd))
    eng.init_zero_state()
    plan = eng.plan(cd)
    timed = plan.layout_info.get("tuned_on_device_ms")
    if plan.layout_info["candidates_by_passes"][plan.layout_info["passes_chosen"]] > 1:
        assert timed and len(timed) >= 2 and all(t > 0 for t in timed)
    psi = eng.state.download()
    assert psi[0] == 1.0 and not np.any(psi[1:]) and eng.l2p is None          # |0..0>, identity layout
    eng.execute(plan)
    np.testing.assert_allclose(eng.state_vector(), want, rtol=0, atol=1e-10)
    before = eng.state.download()
    plan2 = eng.plan(cd)                                                        # a used state: no timing runs
    assert "tuned_on_device_ms" not in plan2.layout_info
    np.testing.assert_array_equal(eng.state.download(), before)
    eng.close()
