"""GPU parity tests proper: the HIP path (through the C ABI) against the CPU oracle and the
reference's golden vectors.  fp64 tolerance: 1e-12 per kernel call (the oracle and the GPU
differ only by FMA contraction / summation order, ~1e-16 relative); circuits 1e-10 as the
north star states."""
import numpy as np
import pytest

from oracle import dense_oracle as orc
from tests.golden_io import golden_circuits, npz

pytestmark = pytest.mark.gpu

ATOL_KERNEL = 1e-12
ATOL_CIRCUIT = 1e-10


def _hip_namespace():
    from quantum_simulations_amd.kernel import gpu_local, gpu_nonlocal, ref_dense
    from quantum_simulations_amd.kernel.device import DeviceChunk, device_count
    assert device_count() >= 1

    class NS:
        pass
    ns = NS()
    ns.local, ns.nonlocal_, ns.ref_dense, ns.DeviceChunk = gpu_local, gpu_nonlocal, ref_dense, DeviceChunk
    return ns


@pytest.fixture(scope="module")
def hip():
    return _hip_namespace()


def _rand_state(n, seed):
    rng = np.random.default_rng(seed)
    v = rng.standard_normal(1 << n) + 1j * rng.standard_normal(1 << n)
    return (v / np.linalg.norm(v)).astype(np.complex128)


def _rand_unitary(dim, seed):
    rng = np.random.default_rng(seed)
    z = rng.standard_normal((dim, dim)) + 1j * rng.standard_normal((dim, dim))
    q, r = np.linalg.qr(z)
    return q * (np.diag(r) / np.abs(np.diag(r)))


# ---------------------------------------------------------------- golden kernels (G3)
def test_golden_apply_1q(hip):
    z = npz("kernels.npz")
    for key in sorted(k for k in z if k.startswith("k1|")):
        _, gname, q = key.split("|")
        chunk = z["chunk_in"].copy()
        hip.local.apply_1q(chunk, int(q[2:]), z[f"m1_{gname}"])
        np.testing.assert_allclose(chunk, z[key], rtol=0, atol=ATOL_KERNEL, err_msg=key)


def test_golden_apply_2q(hip):
    z = npz("kernels.npz")
    for key in sorted(k for k in z if k.startswith("k2|")):
        _, gname, qa, qb = key.split("|")
        chunk = z["chunk_in"].copy()
        hip.local.apply_2q(chunk, int(qa[3:]), int(qb[3:]), z[f"m2_{gname}"])
        np.testing.assert_allclose(chunk, z[key], rtol=0, atol=ATOL_KERNEL, err_msg=key)


def test_golden_nonlocal(hip):
    z = npz("kernels.npz")
    quad = [z[f"nl_in_{i}"] for i in range(4)]
    for name in ("H", "U1"):
        c0, c1 = quad[0].copy(), quad[1].copy()
        hip.nonlocal_.apply_1q_pair(c0, c1, z[f"m1_{name}"])
        np.testing.assert_allclose(c0, z[f"nl|1q_pair|{name}|c0"], rtol=0, atol=ATOL_KERNEL)
        np.testing.assert_allclose(c1, z[f"nl|1q_pair|{name}|c1"], rtol=0, atol=ATOL_KERNEL)
    for name in ("CNOT", "CUG3", "U2", "SWAP", "CR3"):
        U = z[f"m2_{name}"]
        for q in (0, 3, 5):
            for fn, tag in ((hip.nonlocal_.apply_2q_pair_qa_local, "qa_local"),
                            (hip.nonlocal_.apply_2q_pair_qb_local, "qb_local")):
                c0, c1 = quad[0].copy(), quad[1].copy()
                fn(c0, c1, q, U)
                np.testing.assert_allclose(c0, z[f"nl|{tag}|{name}|q={q}|c0"], rtol=0, atol=ATOL_KERNEL)
                np.testing.assert_allclose(c1, z[f"nl|{tag}|{name}|q={q}|c1"], rtol=0, atol=ATOL_KERNEL)
        cs = [c.copy() for c in quad]
        hip.nonlocal_.apply_2q_quad(*cs, U)
        for i, c in enumerate(cs):
            np.testing.assert_allclose(c, z[f"nl|quad|{name}|c{i}"], rtol=0, atol=ATOL_KERNEL)


# ---------------------------------------------------------------- golden circuits (G2)
def test_golden_circuits(hip):
    states = npz("states.npz")
    for name, cd in golden_circuits().items():
        got = hip.ref_dense.simulate(cd)
        np.testing.assert_allclose(got, states[name], rtol=0, atol=ATOL_CIRCUIT, err_msg=name)


# ---------------------------------------------------------------- vs oracle, every qubit
@pytest.mark.parametrize("n", [1, 2, 5, 9, 14])
def test_every_1q_target_vs_oracle(hip, n):
    psi0 = _rand_state(n, 100 + n)
    mats = {"H": orc.gate_matrix("H"), "X": orc.gate_matrix("X"), "T": orc.gate_matrix("T"),
            "Z": orc.gate_matrix("Z"), "RY": orc.gate_matrix("RY", {"theta": 0.77}),
            "U": _rand_unitary(2, n), "D": np.diag(np.exp(1j * np.array([0.3, 1.1])))}
    dev = hip.DeviceChunk.from_numpy(psi0)
    for gname, U in mats.items():
        for q in range(n):
            want = psi0.copy()
            orc.apply_1q(want, q, U)
            dev.upload(psi0)
            dev.apply_1q(q, U)
            np.testing.assert_allclose(dev.download(), want, rtol=0, atol=ATOL_KERNEL,
                                       err_msg=f"{gname} q={q} n={n}")
    dev.close()


@pytest.mark.parametrize("n", [2, 3, 7, 11])
def test_every_2q_pair_vs_oracle(hip, n):
    psi0 = _rand_state(n, 200 + n)
    mats = {k: orc.gate_matrix(k, p) for k, p in
            (("CNOT", {}), ("CZ", {}), ("CY", {}), ("SWAP", {}), ("CR", {"k": 3}),
             ("CU", {"U": orc.gate_matrix("G", {"p": 3}), "exponent": 1}))}
    mats["U"] = _rand_unitary(4, n)
    mats["CTRL_B"] = mats["CNOT"][np.ix_([0, 2, 1, 3], [0, 2, 1, 3])]  # control on qb
    mats["DIAG"] = np.diag(np.exp(1j * np.array([0.1, 0.2, 0.3, 0.4])))
    dev = hip.DeviceChunk.from_numpy(psi0)
    pairs = [(a, b) for a in range(n) for b in range(n) if a != b]
    if n > 7:
        rng = np.random.default_rng(n)
        pairs = [pairs[i] for i in rng.choice(len(pairs), size=40, replace=False)]
    for gname, U in mats.items():
        for qa, qb in pairs:
            want = psi0.copy()
            orc.apply_2q(want, qa, qb, U)
            dev.upload(psi0)
            dev.apply_2q(qa, qb, U)
            np.testing.assert_allclose(dev.download(), want, rtol=0, atol=ATOL_KERNEL,
                                       err_msg=f"{gname} qa={qa} qb={qb} n={n}")
    dev.close()


def test_nonlocal_vs_oracle_all_local_bits(hip):
    k = 7
    chunks = [_rand_state(k, 300 + i) for i in range(4)]
    U4s = {"U": _rand_unitary(4, 9), "CNOT": orc.gate_matrix("CNOT"), "CZ": orc.gate_matrix("CZ"),
           "SWAP": orc.gate_matrix("SWAP"), "CTRL_B": orc.gate_matrix("CNOT")[np.ix_([0, 2, 1, 3], [0, 2, 1, 3])]}
    for name, U in U4s.items():
        for q in range(k):
            for fo, fh in ((orc.apply_2q_pair_qa_local, hip.nonlocal_.apply_2q_pair_qa_local),
                           (orc.apply_2q_pair_qb_local, hip.nonlocal_.apply_2q_pair_qb_local)):
                w0, w1 = chunks[0].copy(), chunks[1].copy()
                fo(w0, w1, q, U)
                g0, g1 = chunks[0].copy(), chunks[1].copy()
                fh(g0, g1, q, U)
                np.testing.assert_allclose(g0, w0, rtol=0, atol=ATOL_KERNEL, err_msg=f"{name} q={q}")
                np.testing.assert_allclose(g1, w1, rtol=0, atol=ATOL_KERNEL, err_msg=f"{name} q={q}")
        want = [c.copy() for c in chunks]
        orc.apply_2q_quad(*want, U)
        got = [c.copy() for c in chunks]
        hip.nonlocal_.apply_2q_quad(*got, U)
        for g, w in zip(got, want):
            np.testing.assert_allclose(g, w, rtol=0, atol=ATOL_KERNEL)
    for U in (orc.gate_matrix("H"), orc.gate_matrix("T"), _rand_unitary(2, 4)):
        w0, w1 = chunks[0].copy(), chunks[1].copy()
        orc.apply_1q_pair(w0, w1, U)
        g0, g1 = chunks[0].copy(), chunks[1].copy()
        hip.nonlocal_.apply_1q_pair(g0, g1, U)
        np.testing.assert_allclose(g0, w0, rtol=0, atol=ATOL_KERNEL)
        np.testing.assert_allclose(g1, w1, rtol=0, atol=ATOL_KERNEL)


# ---------------------------------------------------------------- interface behaviour
def test_non_local_raises(hip):  # reference test_kernel_vs_ref.py:35-43
    chunk = np.zeros(4, dtype=np.complex128)
    chunk[0] = 1.0
    with pytest.raises(NotImplementedError, match="non-local"):
        hip.local.apply_1q(chunk, 2, orc.gate_matrix("H"))
    dev = hip.DeviceChunk.zero_state(2)
    with pytest.raises(NotImplementedError, match="non-local"):
        dev.apply_1q(2, orc.gate_matrix("H"))           # raised by the C ABI itself
    with pytest.raises(NotImplementedError, match="non-local"):
        dev.apply_2q(0, 5, orc.gate_matrix("CNOT"))
    with pytest.raises(ValueError):
        dev.apply_2q(1, 1, orc.gate_matrix("CNOT"))
    dev.close()


def test_complex64_chunk_keeps_dtype(hip):
    rng = np.random.default_rng(3)
    c = (rng.standard_normal(64) + 1j * rng.standard_normal(64)).astype(np.complex64)
    want = c.copy()
    orc.apply_1q(want, 3, orc.gate_matrix("H"))
    hip.local.apply_1q(c, 3, orc.gate_matrix("H"))
    assert c.dtype == np.complex64
    np.testing.assert_allclose(c, want, rtol=0, atol=1e-6)


def test_kat_endianness_bell_ghz(hip):
    psi = hip.ref_dense.simulate({"number_of_qubits": 3, "gates": [{"qubits": [0], "gate": "X"}]})
    assert abs(psi[1] - 1) < 1e-12
    psi = hip.ref_dense.simulate({"number_of_qubits": 4, "gates": [{"qubits": [3], "gate": "X"}]})
    assert abs(psi[8] - 1) < 1e-12
    s2 = 1 / np.sqrt(2)
    psi = hip.ref_dense.simulate(golden_circuits()["bell_2q"])
    np.testing.assert_allclose(psi, [s2, 0, 0, s2], atol=1e-12)


def test_apply_ops_pass_and_norm(hip):
    from quantum_simulations_amd.circuit.fusion import batch_levels
    from quantum_simulations_amd.circuit.io import levelize, validate_circuit_dict
    from quantum_simulations_amd.circuits import random_1q_cx_circuit
    cd = validate_circuit_dict(random_1q_cx_circuit(12, depth=10, seed=4))
    want = orc.simulate(cd)
    dev = hip.DeviceChunk.zero_state(12)
    for p in batch_levels(levelize(cd), 12):
        dev.apply_ops(p["local_ops"])
    np.testing.assert_allclose(dev.download(), want, rtol=0, atol=ATOL_CIRCUIT)
    assert abs(dev.norm2() - 1.0) < 1e-12
    dev.close()


# ---------------------------------------------------------------- full-size properties
def test_ghz_closed_form_on_device_26q(hip):
    from quantum_simulations_amd.circuits import generate_ghz_circuit, generate_ghz_qft
    n = 26
    dev = hip.ref_dense.simulate_on_device(generate_ghz_circuit(n))
    assert dev.max_abs_err_closed_form("ghz", n) < ATOL_CIRCUIT
    dev.close()
    n = 20
    dev = hip.ref_dense.simulate_on_device(generate_ghz_qft(n))
    assert dev.max_abs_err_closed_form("ghz_qft", n) < ATOL_CIRCUIT
    got = dev.download(0, 4096)
    np.testing.assert_allclose(got, orc.ghz_qft_closed_form(n, np.arange(4096)), atol=ATOL_CIRCUIT)
    dev.close()


def test_full_size_28q_unitarity_roundtrip(hip):
    """BASELINE config-2 size: H.H = I and X.X = I on every qubit class, norm preserved,
    random state reproduced to 1e-12 after the round trip (size-independent properties)."""
    n = 28
    dev = hip.DeviceChunk.empty(n)
    dev.init_random(28)
    before = dev.download(0, 1 << 16)
    tail = dev.download((1 << n) - 4096, 4096)
    assert abs(dev.norm2() - 1.0) < 1e-12
    H, CX = orc.gate_matrix("H"), orc.gate_matrix("CNOT")
    for q in (0, 3, 5, 6, 13, 27):
        dev.apply_1q(q, H)
        dev.apply_1q(q, H)
    for qa, qb in ((0, 1), (27, 0), (5, 20), (26, 27)):
        dev.apply_2q(qa, qb, CX)
        dev.apply_2q(qa, qb, CX)
    assert abs(dev.norm2() - 1.0) < 1e-12
    np.testing.assert_allclose(dev.download(0, 1 << 16), before, rtol=0, atol=1e-12)
    np.testing.assert_allclose(dev.download((1 << n) - 4096, 4096), tail, rtol=0, atol=1e-12)
    dev.close()


def test_config2_full_circuit_every_amplitude_vs_c_oracle_28q(hip):
    """BASELINE config 2 at its full size: the complete 28-qubit depth-40 random 1q+CX circuit
    (840 gates, the bench workload) through the fused path, every one of the 2^28 amplitudes
    against the C oracle run on the host cores (about half a minute of CPU work)."""
    import os
    from oracle import c_oracle
    from quantum_simulations_amd.circuit.io import validate_circuit_dict
    from quantum_simulations_amd.circuits import random_1q_cx_circuit
    from quantum_simulations_amd.runner.engine import SingleGpuEngine
    n = 28
    cd = validate_circuit_dict(random_1q_cx_circuit(n, depth=40))
    eng = SingleGpuEngine(n, layout="search")            # line-bit qubits for the pass count + tile-pattern layout
    eng.init_zero_state()
    plan = eng.plan(cd)
    assert plan.layout_info["passes_chosen"] <= plan.layout_info["passes_identity"] == 18
    eng.execute(plan)
    assert eng.last_passes == plan.layout_info["passes_chosen"]              # the library took the named tiles
    assert abs(eng.norm2() - 1.0) < 1e-10
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    c_oracle.set_threads(max(1, min(16, cores)))
    want = c_oracle.simulate(cd)
    worst = 0.0
    step = 1 << 24
    assert eng.l2p is not None and sorted(eng.l2p) == list(range(n))                                 # a layout was chosen
    for off in range(0, 1 << n, step):
        got = eng.state.download(off, step)
        worst = max(worst, float(np.max(np.abs(got - want[eng.logical_index(off, step)]))))         # (the engine's layout undone)
    eng.close()
    assert worst < ATOL_CIRCUIT, worst


# ---------------------------------------------------------------- fused LDS-tile passes
def _random_ops(n, n_ops, seed):
    rng = np.random.default_rng(seed)
    names1 = ["H", "X", "Y", "Z", "S", "T", "RY", "R", "G", "U", "D"]
    names2 = ["CNOT", "CZ", "CY", "SWAP", "CR", "CU", "U4", "CTRL_B", "DIAG4"]
    ops = []
    for i in range(n_ops):
        if n == 1 or rng.random() < 0.55:
            nm = names1[int(rng.integers(len(names1)))]
            q = int(rng.integers(n))
            if nm == "U":
                U = _rand_unitary(2, seed * 1000 + i)
            elif nm == "D":
                U = np.diag(np.exp(1j * rng.uniform(0, 6, 2)))
            else:
                U = orc.gate_matrix(nm, {"theta": float(rng.uniform(0, 6)), "k": int(rng.integers(1, 6)),
                                         "p": int(rng.integers(2, 6))})
            ops.append(([q], U))
        else:
            nm = names2[int(rng.integers(len(names2)))]
            qa, qb = (int(x) for x in rng.choice(n, size=2, replace=False))
            if nm == "U4":
                U = _rand_unitary(4, seed * 1000 + i)
            elif nm == "CTRL_B":
                U = orc.gate_matrix("CY")[np.ix_([0, 2, 1, 3], [0, 2, 1, 3])]
            elif nm == "DIAG4":
                U = np.diag(np.exp(1j * rng.uniform(0, 6, 4)))
            else:
                U = orc.gate_matrix(nm, {"k": int(rng.integers(1, 6)), "U": orc.gate_matrix("G", {"p": 3}),
                                         "exponent": int(rng.integers(1, 3))})
            ops.append(([qa, qb], U))
    return ops


@pytest.mark.parametrize("n", [8, 9, 11, 12, 13, 16, 18])
def test_fused_tile_passes_vs_oracle(hip, n):
    for seed in range(3):
        ops = _random_ops(n, 90, 50 * n + seed)
        psi0 = _rand_state(n, 700 + n + seed)
        want = psi0.copy()
        orc.apply_ops(want, ops)
        dev = hip.DeviceChunk.from_numpy(psi0)
        passes = dev.apply_ops(ops, fused=True)
        got = dev.download()
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-11, err_msg=f"n={n} seed={seed}")
        assert 1 <= passes <= len(ops)
        dev.upload(psi0)
        assert dev.apply_ops(ops, fused=False) == len(ops)
        np.testing.assert_allclose(dev.download(), want, rtol=0, atol=1e-11)
        dev.close()


def test_fused_pass_counts_and_layer_structure(hip):
    """A 1q layer on 20 qubits needs ceil((20-4)/(T-4)) tile passes (T = 11: 3), not 20; CNOT
    controls and CZ/T bits outside the tile ride along as predicates."""
    n = 20
    H, T, CX, CZ = (orc.gate_matrix(g) for g in ("H", "T", "CNOT", "CZ"))
    layer = [([q], H) for q in range(n)]
    dev = hip.DeviceChunk.zero_state(n)
    assert dev.apply_ops(layer) in (2, 3)
    np.testing.assert_allclose(np.abs(dev.download(0, 4096)), 2.0 ** (-n / 2), atol=1e-14)
    diag = [([q], T) for q in range(n)] + [([q, (q + 7) % n], CZ) for q in range(n)]
    assert dev.apply_ops(diag) == 1          # all-diagonal: one pass whatever the qubits
    ctrl = [([19 - q, q], CX) for q in range(7)]   # controls 19..13 outside / targets 0..6
    assert dev.apply_ops(ctrl) == 1
    psi = np.full(1 << 12, 2.0 ** -6, dtype=np.complex128)
    ops12 = [([q], T) for q in range(12)] + [([q, (q + 5) % 12], CZ) for q in range(12)] + \
            [([11 - q, q], CX) for q in range(4)]
    want = psi.copy()
    orc.apply_ops(want, ops12)
    d12 = hip.DeviceChunk.from_numpy(psi)
    d12.apply_ops(ops12)
    np.testing.assert_allclose(d12.download(), want, rtol=0, atol=1e-13)
    d12.close()
    dev.close()


# ---------------------------------------------------------------- full-size closed-form checks
def test_ghz_qft_30q_fused_closed_form(hip):
    """BASELINE config-5 circuit family at one GPU's share (30 qubits, 16 GiB, 495 gates, fused
    passes): every amplitude within 1e-10 of 2^-(n+1)/2 (1 + exp(-2 pi i y / 2^n)), evaluated
    on the device over all 2^30 indices, plus a host-side sample."""
    from quantum_simulations_amd.circuit.io import validate_circuit_dict
    from quantum_simulations_amd.circuits import generate_ghz_qft
    from quantum_simulations_amd.runner.engine import gate_ops
    n = 30
    dev = hip.DeviceChunk.zero_state(n)
    passes = dev.apply_ops(gate_ops(validate_circuit_dict(generate_ghz_qft(n))))
    assert passes < 200
    assert dev.max_abs_err_closed_form("ghz_qft", n) < ATOL_CIRCUIT
    assert abs(dev.norm2() - 1.0) < 1e-12
    for off in (0, (1 << 29) + 12345, (1 << n) - 4096):
        got = dev.download(off, 4096)
        want = orc.ghz_qft_closed_form(n, np.arange(off, off + 4096))
        np.testing.assert_allclose(got, want, rtol=0, atol=ATOL_CIRCUIT)
    dev.close()


def test_ghz_33q_single_gpu_64bit_indexing(hip):
    """33 qubits = 2^33 amplitudes = 128 GiB on ONE MI355X (the size 8 GPUs share in BASELINE
    config 5): GHZ ladder, gate by gate and fused, checked against the closed form on-device;
    exercises 64-bit index arithmetic and 2^23-block grids."""
    from quantum_simulations_amd.circuit.io import validate_circuit_dict
    from quantum_simulations_amd.circuits import generate_ghz_circuit
    from quantum_simulations_amd.runner.engine import gate_ops
    n = 33
    try:
        dev = hip.DeviceChunk.zero_state(n)
    except MemoryError:
        pytest.skip("device has less than 128 GiB free")
    ops = gate_ops(validate_circuit_dict(generate_ghz_circuit(n)))
    dev.apply_ops(ops, fused=False)
    assert dev.max_abs_err_closed_form("ghz", n) < ATOL_CIRCUIT
    assert abs(dev.norm2() - 1.0) < 1e-12
    dev.init_zero(True)
    dev.apply_ops(ops, fused=True)
    assert dev.max_abs_err_closed_form("ghz", n) < ATOL_CIRCUIT
    tail = dev.download((1 << n) - 2, 2)
    assert abs(tail[1] - 2 ** -0.5) < 1e-12 and tail[0] == 0
    dev.close()


def test_config3_sweep_is_unitary_30q(hip):
    """BASELINE config 3 at full size: H on every target 0..29 of a 30-qubit random state, twice
    (H.H = I): the state must come back to 1e-12 and the norm must hold after every gate."""
    n = 30
    dev = hip.DeviceChunk.empty(n)
    dev.init_random(30)
    probes = [(0, 4096), ((1 << 29) - 2048, 4096), ((1 << n) - 4096, 4096)]
    before = [dev.download(o, c) for o, c in probes]
    H = orc.gate_matrix("H")
    for q in range(n):
        dev.apply_1q(q, H)
        if q % 7 == 0:
            assert abs(dev.norm2() - 1.0) < 1e-12
        dev.apply_1q(q, H)
    for (o, c), want in zip(probes, before):
        np.testing.assert_allclose(dev.download(o, c), want, rtol=0, atol=1e-12)
    dev.close()


def test_closed_form_checker_with_staged_layout(hip):
    """The on-device closed-form checker undoes a staging permutation: permute GHZ+QFT amplitudes
    on the host, upload, check with log_to_phys."""
    from quantum_simulations_amd.circuits import generate_ghz_qft
    n = 12
    psi = orc.simulate(generate_ghz_qft(n))
    l2p = list(np.random.default_rng(5).permutation(n))
    x = np.arange(1 << n)
    y = np.zeros_like(x)
    for q, p in enumerate(l2p):
        y |= ((x >> p) & 1) << q
    phys = psi[y]                      # physical index x holds logical amplitude y(x)
    dev = hip.DeviceChunk.from_numpy(phys)
    assert dev.max_abs_err_closed_form("ghz_qft", n, 0, [int(p) for p in l2p]) < 1e-12
    assert dev.max_abs_err_closed_form("ghz_qft", n, 0) > 1e-3      # wrong without the mapping
    dev.close()


def test_config5_ghz_qft_33q_every_amplitude_on_one_gpu(hip):
    """BASELINE config 5 at FULL size on one MI355X (2^33 amplitudes = 128 GiB, 594 gates, fused):
    every amplitude within 1e-10 of the closed form, evaluated on the device."""
    from quantum_simulations_amd.circuit.io import validate_circuit_dict
    from quantum_simulations_amd.circuits import generate_ghz_qft
    from quantum_simulations_amd.runner.engine import gate_ops
    n = 33
    try:
        dev = hip.DeviceChunk.zero_state(n)
    except MemoryError:
        pytest.skip("device has less than 128 GiB free")
    ops = gate_ops(validate_circuit_dict(generate_ghz_qft(n)))
    assert len(ops) == 594
    passes = dev.apply_ops(ops)
    assert passes < 40
    assert dev.max_abs_err_closed_form("ghz_qft", n) < ATOL_CIRCUIT
    assert abs(dev.norm2() - 1.0) < 1e-11
    off = (1 << 32) + 777
    np.testing.assert_allclose(dev.download(off, 2048), orc.ghz_qft_closed_form(n, np.arange(off, off + 2048)),
                               rtol=0, atol=ATOL_CIRCUIT)
    dev.close()


def _streaming_counts(entries):
    return sum(e["launches"] for e in entries), sum(e["streaming_launches"] for e in entries)


def test_streaming_policy_on_small_views_of_a_large_allocation(hip):
    """The cache policy follows the ALLOCATION a chunk lives in (gate_plan.h group_resident, tile_planner.h
    launch_tile): a 2^5..2^16 view of a parent larger than the 256 MiB Infinity Cache runs the non-temporal
    (streaming) instantiations of k_gate / k_gate_shuffle / k_tile, which the stand-alone small-state tests of this
    module never reach -- product-reachable through the chunked runner (views of one big state).  Every 1q target,
    sampled 2q pairs, the four partner-chunk forms, the one-device re-layout and fused passes are checked against
    the oracle (cpu_scalar.py:21-47, cpu_nonlocal.py:22-67 semantics), and the library's launch profile must report
    the streaming instantiation for them (and never for a stand-alone small chunk): forcing `nt` false in the
    launcher fails this test."""
    parent = hip.DeviceChunk.empty(25)                      # 2^25 amplitudes = 512 MiB > 256 MiB
    parent.init_zero(False)
    rng = np.random.default_rng(2025)
    U4s = [_rand_unitary(4, 77), orc.gate_matrix("CNOT"), orc.gate_matrix("CZ"), orc.gate_matrix("SWAP"),
           orc.gate_matrix("CY")[np.ix_([0, 2, 1, 3], [0, 2, 1, 3])], orc.gate_matrix("CR", {"k": 3})]
    for k in range(5, 17):
        n = 1 << k
        offset = int(rng.integers(1, (1 << 25) // n)) * n     # somewhere inside the parent, aligned to the view
        view = parent.view(offset, k)
        psi = _rand_state(k, 9000 + k)
        view.upload(psi)
        want = psi.copy()
        # every 1q target: a dense 2x2 (k_gate<2> above the line bits, k_gate_shuffle<1,1> inside a line)
        view.profile_begin()
        for q in range(k):
            U = _rand_unitary(2, 100 * k + q)
            orc.apply_1q(want, q, U)
            view.apply_1q(q, U)
        launches, streaming = _streaming_counts(view.profile_end())
        assert launches == k and streaming == k, (k, launches, streaming)
        np.testing.assert_allclose(view.download(), want, rtol=0, atol=ATOL_KERNEL, err_msg=f"1q k={k}")
        # diagonal 1q (subset form) and sampled 2q pairs of every classification
        view.profile_begin()
        for q in range(k):
            orc.apply_1q(want, q, orc.gate_matrix("T"))
            view.apply_1q(q, orc.gate_matrix("T"))
        for i in range(12):
            qa, qb = (int(x) for x in rng.choice(k, size=2, replace=False))
            U = U4s[i % len(U4s)]
            orc.apply_2q(want, qa, qb, U)
            view.apply_2q(qa, qb, U)
        launches, streaming = _streaming_counts(view.profile_end())
        # (a removed bit inside a 128-B line keeps plain accesses: two instructions would share a line)
        assert launches == k + 12 and streaming >= k + 6, (k, launches, streaming)
        np.testing.assert_allclose(view.download(), want, rtol=0, atol=ATOL_KERNEL, err_msg=f"2q k={k}")
        if k >= 8:                                            # fused tile passes on the view (k_tile<k or 11, NT>)
            ops = _random_ops(k, 60, 31 * k)
            orc.apply_ops(want, ops)
            view.profile_begin()
            passes = view.apply_ops(ops, fused=True)
            launches, streaming = _streaming_counts(view.profile_end())
            assert launches == passes and streaming == passes, (k, passes, launches, streaming)
            np.testing.assert_allclose(view.download(), want, rtol=0, atol=1e-11, err_msg=f"fused k={k}")
        view.close()
        # partner-chunk forms on four views of the same parent (chunk index = 2 bit(qa) + bit(qb))
        base = int(rng.integers(0, (1 << 25) // (4 * n))) * 4 * n
        views = [parent.view(base + i * n, k) for i in range(4)]
        chunks = [_rand_state(k, 9500 + 4 * k + i) for i in range(4)]
        for v, c in zip(views, chunks):
            v.upload(c)
        views[0].profile_begin()
        U2, U4 = _rand_unitary(2, 5 * k), _rand_unitary(4, 7 * k)
        orc.apply_1q_pair(chunks[0], chunks[1], U2)
        hip.nonlocal_.apply_1q_pair(views[0], views[1], U2)
        for q in sorted({0, 2, 3, k // 2, k - 1}):
            for U in (U4, orc.gate_matrix("CNOT")):
                orc.apply_2q_pair_qa_local(chunks[2], chunks[3], q, U)
                hip.nonlocal_.apply_2q_pair_qa_local(views[2], views[3], q, U)
                orc.apply_2q_pair_qb_local(chunks[0], chunks[2], q, U)
                hip.nonlocal_.apply_2q_pair_qb_local(views[0], views[2], q, U)
        orc.apply_2q_quad(*chunks, U4)
        hip.nonlocal_.apply_2q_quad(*views, U4)
        launches, streaming = _streaming_counts(views[0].profile_end())
        assert launches > 0 and streaming >= launches - 8, (k, launches, streaming)   # (the q < 3 local bits: plain)
        for v, c in zip(views, chunks):
            np.testing.assert_allclose(v.download(), c, rtol=0, atol=ATOL_KERNEL, err_msg=f"partner forms k={k}")
        # one-device re-layout: local bits <-> chunk-index bits (qsim_swap_global_local)
        lb = [int(x) for x in rng.choice(k, size=2, replace=False)]
        full = np.concatenate([v.download() for v in views])     # (bit-exact reference: a re-layout only moves data)
        hip.nonlocal_.swap_global_local(views, [0, 1], lb)
        src = np.arange(4 * n)
        for g, l in zip((k, k + 1), lb):
            bg, bl = (src >> g) & 1, (src >> l) & 1
            src = src ^ ((bg ^ bl) << g) ^ ((bg ^ bl) << l)
        moved = full[src]
        for i, v in enumerate(views):
            np.testing.assert_array_equal(v.download(), moved[i * n:(i + 1) * n], err_msg=f"re-layout k={k}")
            v.close()
    parent.close()
    # control: the same launches on a stand-alone small chunk (fits the Infinity Cache) use plain accesses
    dev = hip.DeviceChunk.from_numpy(_rand_state(12, 1))
    dev.profile_begin()
    for q in range(12):
        dev.apply_1q(q, _rand_unitary(2, q))
    dev.apply_ops(_random_ops(12, 40, 3), fused=True)
    launches, streaming = _streaming_counts(dev.profile_end())
    assert launches > 12 and streaming == 0, (launches, streaming)
    dev.close()


def test_profiles_are_per_stream_and_usable_from_two_host_threads(hip):
    """qsim_profile_begin / end keep one profile per STREAM (it was one per process although handles on
    different streams may be driven from different host threads): two handles on two streams -- the library's own and
    the default stream (`qsim_wrap` with a null stream) -- each profiled from its own thread at the same time, each sees
    exactly its own launches; a second begin on an open stream fails."""
    import threading
    k = 12
    results, errors = {}, []
    own = hip.DeviceChunk.zero_state(k)                      # the library's stream
    mem = hip.DeviceChunk.zero_state(k)
    mem.sync()
    other = hip.DeviceChunk.wrap_pointer(mem.device_ptr, k, 0, stream=0, keep=mem)    # the same kind of memory on the default stream
    barrier = threading.Barrier(2)

    def work(name, c, n_gates):
        try:
            c.profile_begin()
            with pytest.raises(ValueError, match="already open"):
                c.profile_begin()
            barrier.wait(timeout=60)                 # both profiles are open before either launches
            for q in range(n_gates):
                c.apply_1q(q % k, orc.gate_matrix("H"))
            barrier.wait(timeout=60)                 # ... and both have launched before either closes
            prof = c.profile_end()
            results[name] = (sum(e["launches"] for e in prof), c.norm2())
            with pytest.raises(ValueError, match="no profile"):
                c.profile_end()
        except Exception as e:                       # noqa: BLE001 (reported by the main thread)
            errors.append((name, repr(e)))
            try:
                barrier.abort()
            except Exception:
                pass

    threads = [threading.Thread(target=work, args=("a", own, 5)), threading.Thread(target=work, args=("b", other, 9))]
    for t in threads:
        t.start()
    for t in threads:
        t.join(120)
    assert not errors, errors
    assert results["a"][0] == 5 and results["b"][0] == 9, results
    assert abs(results["a"][1] - 1.0) < 1e-12 and abs(results["b"][1] - 1.0) < 1e-12
    for c in (own, other, mem):
        c.close()


@pytest.mark.parametrize("n", [9, 12, 15])
def test_fused_phase_runs_keep_their_order(hip, n):
    """Phase gates that share a predicate are merged per register group (OPC_DIAGR) and written out
    late; a non-diagonal gate on one of their bits must still see them in list order.  Dense in
    phase gates (T, R, CR with few distinct controls) interleaved with H / RY / CNOT on the same
    qubits, against the oracle."""
    for seed in range(4):
        rng = np.random.default_rng(4000 + 10 * n + seed)
        ctrls = [int(q) for q in rng.choice(n, size=3, replace=False)]
        ops = []
        for _ in range(160):
            r = rng.random()
            q = int(rng.integers(n))
            if r < 0.45:
                c = ctrls[int(rng.integers(3))]
                if c == q:
                    continue
                pair = [c, q] if rng.random() < 0.5 else [q, c]
                ops.append((pair, orc.gate_matrix("CR", {"k": int(rng.integers(1, 7))})))
            elif r < 0.65:
                ops.append(([q], orc.gate_matrix(("T", "R", "S", "Z")[int(rng.integers(4))], {"k": int(rng.integers(1, 7))})))
            elif r < 0.85:
                ops.append(([q], orc.gate_matrix(("H", "RY", "X")[int(rng.integers(3))], {"theta": float(rng.uniform(0, 6))})))
            else:
                t = int(rng.integers(n))
                if t != q:
                    ops.append(([q, t], orc.gate_matrix("CNOT")))
        psi0 = _rand_state(n, 900 + seed)
        want = psi0.copy()
        orc.apply_ops(want, ops)
        dev = hip.DeviceChunk.from_numpy(psi0)
        dev.apply_ops(ops, fused=True)
        np.testing.assert_allclose(dev.download(), want, rtol=0, atol=1e-11, err_msg=f"n={n} seed={seed}")
        dev.close()


def test_interleaved_fused_passes_gate_kernels_and_reductions_on_one_stream(hip):
    """The shape of the code that exposed round 2's END_DIRECT bug: several chunks of ONE
    allocation on ONE stream, fused passes (direct-in / direct-out ends, 2 tiles per workgroup) alternating with
    per-gate kernels, partner-chunk butterflies and reductions, nothing synchronised in between except where a
    result is read -- the compiler-visible registers around the engine's asm statement are reused differently by
    every neighbour.  Every amplitude of every chunk against the oracle."""
    k, n_chunks = 20, 4
    parent = hip.DeviceChunk.empty(k + 2)
    views = [parent.view(c << k, k) for c in range(n_chunks)]
    chunks = [_rand_state(k, 1200 + c) for c in range(n_chunks)]
    for v, c in zip(views, chunks):
        v.upload(c)
    rng = np.random.default_rng(77)
    U2 = [_rand_unitary(2, 60 + i) for i in range(4)]
    U4 = _rand_unitary(4, 70)
    for rnd in range(3):
        for c in range(n_chunks):
            ops = _random_ops(k, 70, 9100 + 10 * rnd + c)
            orc.apply_ops(chunks[c], ops)
            assert views[c].apply_ops(ops, fused=True) >= 2
            q = int(rng.integers(k))
            orc.apply_1q(chunks[c], q, U2[c])
            views[c].apply_1q(q, U2[c])                              # per-gate kernel right behind the fused passes
            if c % 2:
                qa, qb = (int(x) for x in rng.choice(k, size=2, replace=False))
                orc.apply_2q(chunks[c], qa, qb, U4)
                views[c].apply_2q(qa, qb, U4)
            else:
                n2 = views[c].norm2()                                # a reduction (reads the chunk, syncs)
                assert abs(n2 - float(np.vdot(chunks[c], chunks[c]).real)) < 1e-10
        # partner-chunk forms across the chunks, then fused passes again on their results
        orc.apply_1q_pair(chunks[0], chunks[1], U2[rnd])
        hip.nonlocal_.apply_1q_pair(views[0], views[1], U2[rnd])
        orc.apply_2q_pair_qb_local(chunks[2], chunks[3], 5 + rnd, U4)
        hip.nonlocal_.apply_2q_pair_qb_local(views[2], views[3], 5 + rnd, U4)
        ops = _random_ops(k, 40, 9500 + rnd)
        for c in (1, 3):
            orc.apply_ops(chunks[c], ops)
            views[c].apply_ops(ops, fused=True)
    for c in range(n_chunks):
        np.testing.assert_allclose(views[c].download(), chunks[c], rtol=0, atol=1e-10, err_msg=f"chunk {c}")
        views[c].close()
    parent.close()


def test_plan_cache_reuses_plans_and_tells_op_lists_apart(hip):
    """qsim_apply_ops keeps the pass images of the last op lists (csrc/tile_planner.h, plan cache): an identical call
    launches them again without planning -- on another chunk, inside a re-layout -- and anything that differs in a
    single matrix entry, qubit or size is planned afresh.  Every result against the oracle."""
    n = 14
    base = _random_ops(n, 80, 4242)
    variants = []
    for i in range(11):                                        # more op lists than the cache holds (8): evictions
        ops = list(base)
        ops[7 * i % len(ops)] = ([i % n], orc.gate_matrix("RY", {"theta": 0.1 + 0.01 * i}))
        variants.append(ops)
    tweaked = list(base)
    qs, U = tweaked[40]
    tweaked[40] = (qs, U * np.exp(1e-9j))                      # the same list but for one matrix, by 1e-9
    moved = [([(q + 1) % n for q in qs], U) for qs, U in base]  # the same matrices on other qubits
    a, b = hip.DeviceChunk.empty(n), hip.DeviceChunk.empty(n)
    for rnd in range(2):
        for idx, ops in enumerate([base] + variants + [base, tweaked, moved, base]):
            psi = _rand_state(n, 1000 * rnd + idx)
            want = psi.copy()
            orc.apply_ops(want, ops)
            dev = a if (rnd + idx) % 2 else b
            dev.upload(psi)
            dev.apply_ops(ops)
            np.testing.assert_allclose(dev.download(), want, rtol=0, atol=1e-11, err_msg=f"round {rnd} list {idx}")
    # a global phase of 1e-9 on one matrix must show (a stale plan would not carry it)
    psi = _rand_state(n, 5)
    a.upload(psi)
    a.apply_ops(base)
    r0 = a.download()
    a.upload(psi)
    a.apply_ops(tweaked)
    assert 1e-13 < float(np.max(np.abs(a.download() - r0))) < 1e-9          # (amplitudes ~ 2^-7, phase 1e-9)
    # the cached plan of `base` inside a fused re-layout, and on a chunk of another size (no hit: planned for 15 qubits)
    send, recv = hip.DeviceChunk.empty(n), hip.DeviceChunk.empty(n)
    idx = np.arange(1 << n)
    pat = ((idx >> 9) & 1) | (((idx >> 12) & 1) << 1)
    want = psi.copy()
    orc.apply_ops(want, base)
    recv.init_zero(False)
    a.upload(psi)
    a.apply_ops_io(base, dst=(send, [9, 12], recv, 1))
    slab = (1 << n) >> 2
    for d in range(4):
        got = (recv if d == 1 else send).download(d * slab, slab)
        np.testing.assert_allclose(got, want[pat == d], rtol=0, atol=1e-11)
    big = hip.DeviceChunk.from_numpy(_rand_state(n + 1, 6))
    want = big.download()
    orc.apply_ops(want, base)
    big.apply_ops(base)
    np.testing.assert_allclose(big.download(), want, rtol=0, atol=1e-11)
    for c in (a, b, send, recv, big):
        c.close()


def test_dense_k_qubit_block_against_the_oracle(hip):
    """qsim_apply_fused_k (v3's fused block as a genuine 2^k x 2^k contraction, parallel_gate_applicator.py:315-385):
    random dense unitaries on 1-6 qubits in random order -- line bits included -- on chunks of 4 to 2^20 amplitudes (below
    2^(k+4) amplitudes: the one-workgroup-per-block form; from there on the matrix cores, k = 5 / 6 with the matrix in LDS), on a
    small view of a 512 MiB parent (the streaming instantiation), the tensor-product block of v3 through it, and the
    argument checks."""
    from quantum_simulations_amd.parallel_gate_applicator import ParallelGateApplicator, tensor_product_single_qubits
    rng = np.random.default_rng(77)

    def unitary(dim):
        return np.linalg.qr(rng.standard_normal((dim, dim)) + 1j * rng.standard_normal((dim, dim)))[0]
    for n in (2, 4, 7, 11, 16, 20):
        psi0 = _rand_state(n, 500 + n)
        dev = hip.DeviceChunk.from_numpy(psi0)
        for k in range(1, min(6, n) + 1):
            for trial in range(6 if n <= 16 else 3):
                qs = [int(q) for q in rng.choice(n, size=k, replace=False)]
                if trial == 0:
                    qs = list(range(k))                      # all inside one 128-byte line (k <= 3) / the lowest bits
                elif trial == 1 and k >= 3:
                    qs = [int(q) for q in rng.permutation(list(range(3)) + [int(x) for x in rng.choice(np.arange(3, n), size=k - 3, replace=False)])] \
                        if n - 3 >= k - 3 else qs            # the whole line + others, the caller's order shuffled
                M = unitary(1 << k)
                want = psi0.copy()
                orc.apply_kq(want, qs, M)
                dev.upload(psi0)
                dev.apply_fused_k(qs, M)
                np.testing.assert_allclose(dev.download(), want, rtol=0, atol=ATOL_KERNEL, err_msg=f"n={n} qubits={qs}")
        dev.close()
    # streaming instantiation: a 2^14 view inside a 512 MiB allocation
    parent = hip.DeviceChunk.empty(25)
    view = parent.view(3 << 14, 14)
    psi0 = _rand_state(14, 9)
    for qs in ([3, 9, 12], [13, 5, 8, 10], [0, 6, 11], [4, 12, 7, 3, 9], [13, 3, 8, 5, 11, 6]):
        M = unitary(1 << len(qs))
        want = psi0.copy()
        orc.apply_kq(want, qs, M)
        view.upload(psi0)
        view.profile_begin()
        view.apply_fused_k(qs, M)
        prof = view.profile_end()
        np.testing.assert_allclose(view.download(), want, rtol=0, atol=ATOL_KERNEL, err_msg=str(qs))
        entry = [e for e in prof if e["kernel"].startswith("k_dense")]
        assert len(entry) == 1 and entry[0]["launches"] == 1 and entry[0]["streaming_launches"] == (1 if min(qs) >= 3 else 0)
    view.close()
    parent.close()
    # v3's tensor-product block through the dense entry == the same gates one by one
    n = 12
    psi0 = _rand_state(n, 3)
    dev = hip.DeviceChunk.from_numpy(psi0)
    qubits = [9, 2, 5]
    mats = {q: unitary(2) for q in qubits}
    ParallelGateApplicator().apply_combined_matrix(dev, sorted(qubits), tensor_product_single_qubits(qubits, mats))
    want = psi0.copy()
    for q in qubits:
        orc.apply_1q(want, q, mats[q])
    np.testing.assert_allclose(dev.download(), want, rtol=0, atol=ATOL_KERNEL)
    with pytest.raises(NotImplementedError, match="non-local"):
        dev.apply_fused_k([0, n, 3], np.eye(8))
    with pytest.raises(ValueError):
        dev.apply_fused_k([0, 1, 1], np.eye(8))
    with pytest.raises(ValueError, match="1 <= k <= 6"):
        dev.apply_fused_k([0, 1, 2, 3, 4, 5, 6], np.eye(128))
    with pytest.raises(ValueError):
        dev.apply_fused_k([0, 1, 2], np.eye(4))
    dev.close()
