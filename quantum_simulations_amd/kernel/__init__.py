"""Gate table + HIP-backed butterfly kernels (mirror of wenbo_engine.kernel)."""
