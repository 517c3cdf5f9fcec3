#!/bin/bash
# pass builder with / without commutation-aware blocking (QSIM_PLAN_COMMUTE, probe build), same device
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
for n in 26 28 30; do
  for c in 0 1; do
    printf "n=%s QSIM_PLAN_COMMUTE=%s  " $n $c
    QSIM_PLAN_COMMUTE=$c python3 tools/step_times.py $n 20260228 1 2 3 | tail -1
  done
done
python3 - <<'PY'
import os, sys, time
sys.path.insert(0, ".")
from quantum_simulations_amd.circuits import random_clifford_t_circuit
for c in ("0", "1"):
    pass
PY
