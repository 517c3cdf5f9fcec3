"""The qubit layout chosen for the DRAM pattern of the tiles (runner/tile_layout.py, runner/engine.py): host logic on the
CPU -- the cost model loads and prices tile-bit sets, the search never returns a worse layout than the identity and
returns a permutation that keeps the line bits, the SWAP list between two layouts is right, and a plan written for a
layout names the same passes on the new bits (qsim_plan_ops_tiled: the library takes the named tiles)."""
import ctypes as C

import numpy as np
import pytest

from oracle import dense_oracle as orc
from quantum_simulations_amd.circuit.staging import permute_state
from quantum_simulations_amd.runner import tile_layout
from quantum_simulations_amd.runner.engine import _planned_tile_masks, layout_swaps


def test_cost_model_loads_and_prices_tiles():
    for n in (24, 28, 30, 33):
        m = tile_layout.model_for(n)
        assert m["bit"].shape[0] == m["top"] - 2 and m["pair"].shape == (m["top"] - 2, m["top"] - 2)
        good = tile_layout.tile_cost(m, [3, 4, 5, 6, 12, 13, 14, 15])
        bad = tile_layout.tile_cost(m, [3, 4, 5, 6, 20, 21, 22, 23])
        assert 0.5 * m["c0"] < good < bad, (n, good, bad)          # the fast and the slow set of DESIGN section 3
        assert tile_layout.tile_cost(m, [3, 4, 5, 6, 7, 8, 9, 40]) > 0       # bits above the fitted range are clamped


def test_choose_layout_is_a_permutation_that_keeps_the_line_bits_and_never_loses():
    rng = np.random.default_rng(3)
    for n in (26, 28, 30):
        tiles = [sorted(int(b) for b in rng.choice(np.arange(3, n), size=8, replace=False)) for _ in range(12)]
        l2p, c0, c1 = tile_layout.choose_layout(tiles, n, sweeps=10)
        assert sorted(l2p) == list(range(n)) and l2p[:3] == [0, 1, 2]
        assert c1 <= c0 + 1e-9
        m = tile_layout.model_for(n)
        assert abs(sum(tile_layout.tile_cost(m, [l2p[q] for q in t]) for t in tiles) - c1) < 1e-9
        assert abs(sum(tile_layout.tile_cost(m, t) for t in tiles) - c0) < 1e-9
    assert tile_layout.choose_layout([], 28)[0] == list(range(28))


def test_layout_swaps_move_a_state_between_layouts():
    n = 7
    rng = np.random.default_rng(5)
    psi = rng.standard_normal(1 << n) + 1j * rng.standard_normal(1 << n)
    SW = orc.gate_matrix("SWAP")
    for _ in range(20):
        cur = [int(x) for x in rng.permutation(n)]
        want = [int(x) for x in rng.permutation(n)] if rng.random() < 0.8 else None
        x = np.arange(1 << n)
        y = np.zeros_like(x)
        for q, p in enumerate(cur):
            y |= ((x >> p) & 1) << q
        phys = psi[y]                                     # the state held in layout `cur`
        np.testing.assert_array_equal(permute_state(phys, cur), psi)
        swaps = layout_swaps(cur, want, n)
        assert len(swaps) <= n - 1
        for a, b in swaps:
            orc.apply_2q(phys, a, b, SW)
        np.testing.assert_allclose(permute_state(phys, want if want is not None else list(range(n))), psi, atol=0)
    assert layout_swaps(None, None, n) == [] and layout_swaps([2, 0, 1], [2, 0, 1], 3) == []


def test_named_tiles_give_the_same_passes_on_other_index_bits():
    """qsim_plan_ops_tiled: plan a circuit, move its qubits to other index bits, name the moved tiles: the library plans
    the same number of passes with exactly those tiles, and the passes compute the relabelled circuit (interpreted on the
    CPU against the oracle)."""
    from quantum_simulations_amd import _lib
    from quantum_simulations_amd.kernel.device import pack_ops
    from tests import tile_interpreter as ti
    from tests.test_gpu_kernels import _rand_state, _random_ops
    n = 14
    rng = np.random.default_rng(9)
    ops = _random_ops(n, 150, 4321)
    masks = _planned_tile_masks(n, ops)
    assert len(masks) >= 3
    l2p = [0, 1, 2] + [int(x) for x in 3 + rng.permutation(n - 3)]
    moved_ops = [([l2p[q] for q in qs], U) for qs, U in ops]
    moved_masks = np.array([sum(1 << l2p[b] for b in range(n) if (int(m) >> b) & 1) for m in masks], dtype=np.uint64)
    nq, qubits, mats = pack_ops(moved_ops)
    lib = _lib.load()
    count = C.c_int32()
    args = (n, len(nq), nq.ctypes.data_as(C.c_void_p), qubits.ctypes.data_as(C.c_void_p), mats.ctypes.data_as(C.c_void_p),
            len(moved_masks), moved_masks.ctypes.data_as(C.c_void_p))
    _lib.check(lib.qsim_plan_ops_tiled(*args, None, 0, C.byref(count)))
    assert count.value == len(masks)
    images = np.zeros(count.value, dtype=ti._IMAGE)
    _lib.check(lib.qsim_plan_ops_tiled(*args, images.ctypes.data_as(C.c_void_p), images.nbytes, C.byref(count)))
    for img, want in zip(images, moved_masks):
        assert sum(1 << int(b) for b in img["h"][:int(img["T"]) - 3]) == int(want)
    psi = _rand_state(n, 17)
    want_state = psi.copy()
    orc.apply_ops(want_state, moved_ops)
    ti.run(psi, images)
    np.testing.assert_allclose(psi, want_state, rtol=0, atol=1e-12)
    # a mask that holds no op is ignored: the search takes over and the list still plans
    junk = np.array([1 << 3], dtype=np.uint64)
    _lib.check(lib.qsim_plan_ops_tiled(n, len(nq), nq.ctypes.data_as(C.c_void_p), qubits.ctypes.data_as(C.c_void_p),
                                       mats.ctypes.data_as(C.c_void_p), 1, junk.ctypes.data_as(C.c_void_p), None, 0, C.byref(count)))
    assert count.value >= 1


def test_choose_plan_layout_names_the_planned_passes_and_never_needs_more_of_them():
    """runner/engine.choose_plan_layout on the CPU: the chosen layout is a permutation, needs no more passes than the
    identity (the identity is one of the candidates), and the tile masks it names ARE the passes the library plans for the
    relabelled op list (qsim_plan_ops_tiled takes them: same count, same tiles)."""
    import ctypes as C

    from quantum_simulations_amd import _lib
    from quantum_simulations_amd.circuit.fusion import batch_levels
    from quantum_simulations_amd.circuit.io import levelize, validate_circuit_dict
    from quantum_simulations_amd.circuits import random_1q_cx_circuit
    from quantum_simulations_amd.kernel.device import pack_ops
    from quantum_simulations_amd.runner.engine import choose_plan_layout
    from tests import tile_interpreter as ti
    n = 26
    cd = validate_circuit_dict(random_1q_cx_circuit(n, depth=12, seed=7))
    batches = [p["local_ops"] for p in batch_levels(levelize(cd), n)]
    l2p, masks, info = choose_plan_layout(n, batches, n_candidates=24)
    assert sorted(l2p) == list(range(n))
    assert info["passes_chosen"] <= info["passes_identity"] and info["candidates"] == 25
    assert sum(info["candidates_by_passes"].values()) == 25 and info["model_ms"][1] <= info["model_ms"][0] + 1e-9
    assert sum(len(m) for m in masks) == info["passes_chosen"]
    lib = _lib.load()
    for ops, ms in zip(batches, masks):
        nq, qubits, mats = pack_ops([([l2p[q] for q in qs], U) for qs, U in ops])
        count = C.c_int32()
        args = (n, len(nq), nq.ctypes.data_as(C.c_void_p), qubits.ctypes.data_as(C.c_void_p), mats.ctypes.data_as(C.c_void_p),
                len(ms), ms.ctypes.data_as(C.c_void_p))
        _lib.check(lib.qsim_plan_ops_tiled(*args, None, 0, C.byref(count)))
        assert count.value == len(ms)
        images = np.zeros(count.value, dtype=ti._IMAGE)
        _lib.check(lib.qsim_plan_ops_tiled(*args, images.ctypes.data_as(C.c_void_p), images.nbytes, C.byref(count)))
        for img, want in zip(images, ms):
            assert sum(1 << int(b) for b in img["h"][:int(img["T"]) - 3]) == int(want)


def test_plan_count_layouts_matches_single_plans_and_rejects_bad_layouts():
    import ctypes as C

    from quantum_simulations_amd import _lib
    from quantum_simulations_amd.kernel.device import pack_ops
    from quantum_simulations_amd.runner.engine import _count_passes
    from tests import tile_interpreter as ti
    from tests.test_gpu_kernels import _random_ops
    n = 15
    ops = _random_ops(n, 200, 99)
    rng = np.random.default_rng(2)
    layouts = np.array([np.arange(n)] + [rng.permutation(n) for _ in range(9)], dtype=np.int32)
    counts = _count_passes(n, [ops], layouts, 4)
    for lay, c in zip(layouts, counts):
        assert c == len(ti.plan(n, [([int(lay[q]) for q in qs], U) for qs, U in ops]))
    nq, qubits, mats = pack_ops(ops)
    bad = np.zeros((1, n), dtype=np.int32)
    out = np.zeros(1, dtype=np.int32)
    rc = _lib.load().qsim_plan_count_layouts(n, len(nq), nq.ctypes.data_as(C.c_void_p), qubits.ctypes.data_as(C.c_void_p),
                                             mats.ctypes.data_as(C.c_void_p), 1, bad.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), 2)
    assert rc == -1
