"""Stand-in for qiskit.QuantumCircuit (qiskit is not installed): just the attributes that
`qiskit_to_dict` reads -- num_qubits, data[i].operation.{name,params}, data[i].qubits,
find_bit(q).index."""
from types import SimpleNamespace


class FakeCircuit:
    def __init__(self, num_qubits: int, program):
        self.num_qubits = num_qubits
        self._bits = [object() for _ in range(num_qubits)]
        self.data = [SimpleNamespace(operation=SimpleNamespace(name=name, params=list(params)),
                                     qubits=[self._bits[q] for q in qubits])
                     for name, qubits, params in program]

    def find_bit(self, bit):
        return SimpleNamespace(index=self._bits.index(bit))
