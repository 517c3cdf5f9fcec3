"""Qiskit front-end against G9: the reference's qiskit_to_dict run on the same
duck-typed circuit (tests/fake_qiskit.py; qiskit itself is absent here and on the GPU box)."""
import numpy as np
import pytest

from oracle import dense_oracle as orc
from quantum_simulations_amd.circuit.import_qiskit import SUPPORTED_BASIS, qiskit_to_dict
from quantum_simulations_amd.circuit.io import validate_circuit_dict
from tests.fake_qiskit import FakeCircuit
from tests.golden_io import jdoc


def test_matches_reference_output():
    g = jdoc("qiskit_import.json")
    program = [(name, qubits, params) for name, qubits, params in g["program"]]
    assert qiskit_to_dict(FakeCircuit(g["n_qubits"], program)) == g["expected"]
    assert SUPPORTED_BASIS == g["supported_basis"]
    validate_circuit_dict(g["expected"])


def test_unsupported_gate_message():
    g = jdoc("qiskit_import.json")
    with pytest.raises(ValueError) as e:
        qiskit_to_dict(FakeCircuit(2, [("rz", [0], [0.1])]))
    assert str(e.value) == g["unsupported_message"]


def test_imported_bell_simulates():   # test_qiskit_oracle.py::TestQiskitDirect::test_bell, minus qiskit
    cd = qiskit_to_dict(FakeCircuit(2, [("h", [0], []), ("cx", [0, 1], [])]))
    s = 1 / np.sqrt(2)
    np.testing.assert_allclose(orc.simulate(cd), [s, 0, 0, s], atol=1e-12)
