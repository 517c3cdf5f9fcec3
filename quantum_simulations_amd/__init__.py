"""MI355X-native statevector gate engine (drop-in for the gate-application hot path
of onofreiandrea/quantum_simulations: wenbo_engine/kernel + the v3 amplitude worker).

Python host code -> ctypes -> `libqsim_hip.so` (hand-written gfx950 HIP kernels).
There is no CPU fallback: importing `quantum_simulations_amd.kernel.gpu_local` (or
anything that applies gates) without the built library raises.
"""
__version__ = "0.1.0"
