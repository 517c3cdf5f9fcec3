"""Pin the C oracle (oracle/qsim_oracle.c) and the v1-SQL restatement to the reference's
golden vectors.  CPU only."""
import numpy as np
import pytest

from oracle import c_oracle, dense_oracle, v1_sql_oracle
from tests.golden_io import circuit_from_json, golden_circuits, jdoc, npz


def test_c_oracle_kernels_vs_golden():
    z = npz("kernels.npz")
    for key in sorted(k for k in z if k.startswith("k1|")):
        _, g, q = key.split("|")
        chunk = z["chunk_in"].copy()
        c_oracle.apply_1q(chunk, int(q[2:]), z[f"m1_{g}"])
        np.testing.assert_allclose(chunk, z[key], rtol=0, atol=2e-16, err_msg=key)
    for key in sorted(k for k in z if k.startswith("k2|")):
        _, g, qa, qb = key.split("|")
        chunk = z["chunk_in"].copy()
        c_oracle.apply_2q(chunk, int(qa[3:]), int(qb[3:]), z[f"m2_{g}"])
        np.testing.assert_allclose(chunk, z[key], rtol=0, atol=4e-16, err_msg=key)
    c0, c1 = z["nl_in_0"].copy(), z["nl_in_1"].copy()
    c_oracle.apply_1q_pair(c0, c1, z["m1_U1"])
    np.testing.assert_allclose(c0, z["nl|1q_pair|U1|c0"], rtol=0, atol=1e-15)
    np.testing.assert_allclose(c1, z["nl|1q_pair|U1|c1"], rtol=0, atol=1e-15)


@pytest.mark.parametrize("name", sorted(npz("states.npz")))
def test_c_oracle_circuits_vs_golden(name):
    got = c_oracle.simulate(golden_circuits()[name])
    np.testing.assert_allclose(got, npz("states.npz")[name], rtol=0, atol=1e-14)


def test_c_oracle_nonlocal_code():
    psi = np.zeros(4, dtype=np.complex128)
    with pytest.raises(NotImplementedError, match="non-local"):
        c_oracle.apply_1q(psi, 2, dense_oracle.gate_matrix("H"))


@pytest.mark.parametrize("name", sorted(jdoc("v1_sql_circuits.json")))
def test_v1_sql_restatement_vs_reference_v1(name):
    cd = circuit_from_json(jdoc("v1_sql_circuits.json")[name])
    psi, counts = v1_sql_oracle.run_circuit(cd, return_row_counts=True)
    z = npz("v1_sql.npz")
    np.testing.assert_allclose(psi, z[f"{name}|state"], rtol=0, atol=1e-15)
    assert counts == list(z[f"{name}|rows"])


def test_v1_sql_ghz_row_growth():
    """BASELINE.md 3.1 pins: row count after gate g is 2^g for GHZ; amplitudes exact."""
    n = 10
    gates = [{"qubits": [0], "gate": "H"}] + [{"qubits": [q - 1, q], "gate": "CNOT"} for q in range(1, n)]
    psi, counts = v1_sql_oracle.run_circuit({"number_of_qubits": n, "gates": gates}, True)
    assert counts == [2 ** g for g in range(n + 1)]
    assert psi[0] == psi[-1] == 0.7071067811865475 and np.count_nonzero(psi) == 2
