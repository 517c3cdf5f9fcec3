"""Pin the CPU oracle (oracle/dense_oracle.py) to the reference: golden vectors made by
running the reference in the build container + the reference's own known-answer tests."""
import numpy as np
import pytest

from oracle import dense_oracle as orc
from tests.golden_io import golden_circuits, npz

S2 = 1.0 / np.sqrt(2.0)


def _gate_case(key):
    parts = key.split("|")
    name, params = parts[0], {}
    for kv in parts[1:]:
        k, v = kv.split("=")
        if k == "theta":
            params[k] = {"pi3": np.pi / 3, "pi4": np.pi / 4, "1p234": 1.234}[v]
        elif k == "U":
            params[k] = {"Z": orc.gate_matrix("Z"), "G3": orc.gate_matrix("G", {"p": 3})}[v]
        elif k == "e":
            params["exponent"] = int(v)
        else:
            params[k] = int(v)
    return name, params


@pytest.mark.parametrize("key", sorted(npz("gate_matrices.npz")))
def test_gate_matrices_bit_exact(key):
    name, params = _gate_case(key)
    got = orc.gate_matrix(name, params)
    assert got.dtype == np.complex128
    np.testing.assert_array_equal(got, npz("gate_matrices.npz")[key])


@pytest.mark.parametrize("name", sorted(npz("states.npz")))
def test_simulate_matches_reference(name):
    got = orc.simulate(golden_circuits()[name])
    np.testing.assert_allclose(got, npz("states.npz")[name], rtol=0, atol=1e-15)


def _kernel_keys(prefix):
    return sorted(k for k in npz("kernels.npz") if k.startswith(prefix))


@pytest.mark.parametrize("key", _kernel_keys("k1|"))
def test_apply_1q_bit_exact(key):
    z = npz("kernels.npz")
    _, gname, q = key.split("|")
    chunk = z["chunk_in"].copy()
    orc.apply_1q(chunk, int(q[2:]), z[f"m1_{gname}"])
    np.testing.assert_array_equal(chunk, z[key])


@pytest.mark.parametrize("key", _kernel_keys("k2|"))
def test_apply_2q(key):
    z = npz("kernels.npz")
    _, gname, qa, qb = key.split("|")
    chunk = z["chunk_in"].copy()
    orc.apply_2q(chunk, int(qa[3:]), int(qb[3:]), z[f"m2_{gname}"])
    np.testing.assert_allclose(chunk, z[key], rtol=0, atol=2e-16)


def test_nonlocal_butterflies():
    z = npz("kernels.npz")
    quad = [z[f"nl_in_{i}"] for i in range(4)]
    for name in ("H", "U1"):
        c0, c1 = quad[0].copy(), quad[1].copy()
        orc.apply_1q_pair(c0, c1, z[f"m1_{name}"])
        np.testing.assert_array_equal(c0, z[f"nl|1q_pair|{name}|c0"])
        np.testing.assert_array_equal(c1, z[f"nl|1q_pair|{name}|c1"])
    for name in ("CNOT", "CUG3", "U2", "SWAP", "CR3"):
        U = z[f"m2_{name}"]
        for q in (0, 3, 5):
            for fn, tag in ((orc.apply_2q_pair_qa_local, "qa_local"),
                            (orc.apply_2q_pair_qb_local, "qb_local")):
                c0, c1 = quad[0].copy(), quad[1].copy()
                fn(c0, c1, q, U)
                np.testing.assert_allclose(c0, z[f"nl|{tag}|{name}|q={q}|c0"], rtol=0, atol=2e-16)
                np.testing.assert_allclose(c1, z[f"nl|{tag}|{name}|q={q}|c1"], rtol=0, atol=2e-16)
        cs = [c.copy() for c in quad]
        orc.apply_2q_quad(*cs, U)
        for i, c in enumerate(cs):
            np.testing.assert_allclose(c, z[f"nl|quad|{name}|c{i}"], rtol=0, atol=2e-16)


# ---- the reference's own known-answer tests, restated against the oracle ----------
def test_kat_endianness_lock():  # test_endianness_lock.py:15-23
    psi = orc.simulate({"number_of_qubits": 3, "gates": [{"qubits": [0], "gate": "X"}]})
    assert abs(psi[1]) > 0.999 and np.all(np.abs(np.delete(psi, 1)) < 1e-12)


def test_kat_bell_ghz3_hwall():  # test_ref_known_states.py:10-39
    bell = orc.simulate({"number_of_qubits": 2, "gates": [
        {"qubits": [0], "gate": "H"}, {"qubits": [0, 1], "gate": "CNOT"}]})
    np.testing.assert_allclose(bell, [S2, 0, 0, S2], atol=1e-12)
    g3 = orc.simulate({"number_of_qubits": 3, "gates": [
        {"qubits": [0], "gate": "H"}, {"qubits": [0, 1], "gate": "CNOT"},
        {"qubits": [1, 2], "gate": "CNOT"}]})
    assert abs(g3[0] - S2) < 1e-12 and abs(g3[7] - S2) < 1e-12
    wall = orc.simulate({"number_of_qubits": 4,
                         "gates": [{"qubits": [i], "gate": "H"} for i in range(4)]})
    np.testing.assert_allclose(np.abs(wall), 0.25, atol=1e-12)


def test_kat_x_on_q3_index_8():  # test_nonlocal.py:45-50
    psi = orc.simulate({"number_of_qubits": 4, "gates": [{"qubits": [3], "gate": "X"}]})
    assert abs(psi[8] - 1.0) < 1e-12


def test_non_local_raises():  # test_kernel_vs_ref.py:35-43
    chunk = np.zeros(4, dtype=np.complex128)
    with pytest.raises(NotImplementedError, match="non-local"):
        orc.apply_1q(chunk, 2, orc.gate_matrix("H"))


@pytest.mark.parametrize("n", [3, 6, 10])
def test_ghz_qft_closed_form(n):  # SURVEY 8c known answer for config 5
    ref = npz("states.npz")[f"v1_ghz_qft_{n}"]
    np.testing.assert_allclose(orc.ghz_qft_closed_form(n, np.arange(1 << n)), ref, atol=1e-14)


def test_permute_state_matches_reference():
    from tests.golden_io import c128, jdoc
    for case in jdoc("planner.json")["permute"]:
        got = orc.permute_state(c128(case["in"]), case["log_to_phys"])
        np.testing.assert_array_equal(got, c128(case["out"]))


def test_dense_k_qubit_block_is_pinned_through_the_pinned_kernels():
    """oracle.apply_kq restates v3's `_apply_combined_matrix` (Spark cannot run here): a Kronecker product of 1q matrices
    must equal the pinned `apply_1q` on every qubit (that IS v3's fused block, parallel_gate_applicator.py:169-204), and a
    4x4 the pinned `apply_2q` with qubits[1] as the pair's MSB."""
    from oracle import dense_oracle as orc
    rng = np.random.default_rng(8)
    n = 9
    psi0 = rng.standard_normal(1 << n) + 1j * rng.standard_normal(1 << n)
    for k in (1, 2, 3, 4):
        for _ in range(6):
            qs = [int(q) for q in rng.choice(n, size=k, replace=False)]
            mats = [np.linalg.qr(rng.standard_normal((2, 2)) + 1j * rng.standard_normal((2, 2)))[0] for _ in qs]
            M = np.array([[1.0]], dtype=complex)
            for U in mats:                                   # pattern bit i <-> qubits[i]: later factors are more significant
                M = np.kron(U, M)
            a, b = psi0.copy(), psi0.copy()
            orc.apply_kq(a, qs, M)
            for q, U in zip(qs, mats):
                orc.apply_1q(b, q, U)
            np.testing.assert_allclose(a, b, rtol=0, atol=1e-13)
    for _ in range(6):
        qs = [int(q) for q in rng.choice(n, size=2, replace=False)]
        U = np.linalg.qr(rng.standard_normal((4, 4)) + 1j * rng.standard_normal((4, 4)))[0]
        a, b = psi0.copy(), psi0.copy()
        orc.apply_kq(a, qs, U)
        orc.apply_2q(b, qs[1], qs[0], U)
        np.testing.assert_allclose(a, b, rtol=0, atol=1e-13)
    with pytest.raises(NotImplementedError, match="non-local"):
        orc.apply_kq(psi0.copy(), [0, n], np.eye(4))
    with pytest.raises(ValueError):
        orc.apply_kq(psi0.copy(), [0, 1], np.eye(8))
