import time, numpy as np, sys
sys.path.insert(0,'.')
from quantum_simulations_amd.kernel import gpu_local, gates
n=26
psi=np.zeros(1<<n,dtype=np.complex128); psi[0]=1
gpu_local.apply_1q(psi, 3, gates.H())
t0=time.perf_counter()
for q in (0,10,25): gpu_local.apply_1q(psi, q, gates.H())
dt=(time.perf_counter()-t0)/3
print(f"host-buffer apply_1q n={n}: {dt*1e3:.1f} ms per call = {32*(1<<n)/dt/1e9:.2f} GB/s algorithmic (PCIe both ways included)")
