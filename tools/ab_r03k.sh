#!/bin/bash
# A/B on one device: swap instruction (v_swap_b32 vs v_mov_b64 triples) and ops-per-pass caps
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
P=$R/quantum_simulations_amd
for rep in 1 2; do
for n in 24 28; do
  for lib in libqsim_hip_probes.so libqsim_hip_mov64.so; do
    printf "n=%s %-26s " $n $lib
    QSIM_LIBRARY=$P/$lib python3 tools/step_times.py $n 20260228 1 2 3 | tail -1
  done
done
done
for n in 28; do
  for cap in 128 64 58 54 50 46; do
    printf "n=%s QSIM_PASS_GATES=%-4s " $n $cap
    QSIM_PASS_GATES=$cap python3 tools/step_times.py $n 20260228 1 2 3 | tail -1
  done
done
