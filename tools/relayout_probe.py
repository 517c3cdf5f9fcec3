#!/usr/bin/env python3
"""HBM rate of the re-layout helpers on one GPU: slab pack / unpack (the kernels around the RCCL
all-to-all) for several local bit choices, and the single-device all-to-all among chunks
(qsim_swap_global_local).    python tools/relayout_probe.py [n_qubits]"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from quantum_simulations_amd.kernel import gpu_nonlocal  # noqa: E402
from quantum_simulations_amd.kernel.device import DeviceChunk  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 28
state = DeviceChunk.empty(n)
state.init_random(1)
buf = DeviceChunk.empty(n)
for bits in ([n - 1], [n - 3, n - 2, n - 1], [3, 4, 5], [0, 1, 2], [0, 10, 20], [5, 6, 7]):
    m = len(bits)
    slab = 1 << (n - m)
    for what in ("pack", "unpack"):
        state.sync()
        state.time_begin()
        for pattern in range(1 << m):
            if what == "pack":
                state.pack_bits(bits, pattern, buf, pattern * slab)
            else:
                state.unpack_bits(bits, pattern, buf, pattern * slab)
        ms = state.time_end()
        gb = 32.0 * (1 << n) / 1e9      # every amplitude read once and written once
        print(f"n={n} bits={bits} {what}: {ms:.3f} ms for the whole shard = {gb / (ms * 1e-3):.0f} GB/s (r+w)")
    for what in ("pack_all", "unpack_all"):
        state.sync()
        state.time_begin()
        getattr(state, what)(bits, buf, 0)
        ms = state.time_end()
        moved = 32.0 * (1 << n) * (1 - 0.5 ** m) / 1e9
        print(f"n={n} bits={bits} {what} (one slab stays): {ms:.3f} ms = {moved / (ms * 1e-3):.0f} GB/s (r+w)")
k = n - 3
chunks = [state.view(c << k, k) for c in range(8)]
for lo in ([k - 3, k - 2, k - 1], [0, 1, 2], [3, 10, 17]):
    state.sync()
    state.time_begin()
    gpu_nonlocal.swap_global_local(chunks, [0, 1, 2], lo)
    ms = state.time_end()
    print(f"swap_global_local 8 chunks of 2^{k}, local bits {lo}: {ms:.3f} ms = {32.0 * (1 << n) * 7 / 8 / 1e9 / (ms * 1e-3):.0f} GB/s (r+w of the 7/8 that move)")
