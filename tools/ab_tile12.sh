#!/bin/bash
# 11-bit vs 12-bit tiles with the round-3 pass builder (libqsim_hip.so vs libqsim_hip_tile12.so: `make -C quantum_simulations_amd/csrc tile12`)
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
for n in 28 30; do
  for lib in libqsim_hip.so libqsim_hip_tile12.so; do
    printf "n=%s %-24s " $n $lib
    QSIM_LIBRARY=$R/quantum_simulations_amd/$lib python3 tools/step_times.py $n 20260228 1 2 3 | tail -1
  done
done
