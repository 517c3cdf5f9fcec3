"""qsim_fingerprint (include/qsim_hip.h) against its numpy restatement: identity and staged layouts, shard bases, index-set
filters, argument checks."""
import numpy as np
import pytest

from tests.cpu_shard_backend import fingerprint_np

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from quantum_simulations_amd.kernel.device import DeviceChunk
    return DeviceChunk


@pytest.mark.parametrize("k", [3, 9, 14, 20])
def test_fingerprint_matches_numpy(hip, k):
    rng = np.random.default_rng(100 + k)
    psi = rng.standard_normal(1 << k) + 1j * rng.standard_normal(1 << k)
    psi /= np.linalg.norm(psi)
    c = hip.empty(k, 0)
    c.upload(psi)
    for n_total, base_rank in ((k, 0), (k + 2, 3), (k + 3, 5)):
        base = base_rank << k
        l2p = [int(x) for x in rng.permutation(n_total)]
        for perm in (None, l2p):
            for seed in (0, 12345):
                got = c.fingerprint(n_total, base, perm, seed)
                want = fingerprint_np(psi, n_total, base, perm, seed)
                assert abs(got - want) < 1e-13, (k, n_total, perm is not None, seed)
            mask = int(rng.integers(1, 1 << n_total))
            value = int(rng.integers(0, 1 << n_total)) & mask
            got = c.fingerprint(n_total, base, perm, 9, mask, value)
            assert abs(got - fingerprint_np(psi, n_total, base, perm, 9, mask, value)) < 1e-13
    # the filters of all values of a mask partition the sum
    whole = c.fingerprint(k, 0, None, 4)
    parts = sum(c.fingerprint(k, 0, None, 4, 0b101, v) for v in (0, 1, 4, 5))
    assert abs(whole - parts) < 1e-13
    c.close()


def test_fingerprint_rejects_bad_arguments(hip):
    c = hip.empty(6, 0)
    c.init_zero(True)
    with pytest.raises(ValueError):
        c.fingerprint(5)                       # fewer total qubits than the chunk holds
    with pytest.raises(ValueError):
        c.fingerprint(8, 13)                   # base not a multiple of the chunk length
    with pytest.raises(ValueError):
        c.fingerprint(8, 4 << 6)               # the chunk does not fit the state at that base
    with pytest.raises(ValueError):
        c.fingerprint(8, 0, [0, 1, 2, 3, 4, 5, 6, 6])
    with pytest.raises(ValueError):
        c.fingerprint(8, 0, None, 0, 0b1, 0b10)
    assert abs(c.fingerprint(6, 0, None, 3) - fingerprint_np(np.eye(1, 64, 0, dtype=np.complex128)[0], 6, 0, None, 3)) < 1e-15
    c.close()
