#!/bin/bash
# What does a fused pass cost WITH its gates when the tile always holds the low index bits 3..3+N-1 (2^(N+7)-byte
# contiguous pieces)?  Probe build, QSIM_PLAN_FORCE_LOW = 0 / 2 / 3 / 4, bench workload family at 28 (and 30) qubits.
set -e
out=gpurun_out/r04a_anchor_cost.txt
: > $out
for n in 28 30; do
  for low in 0 2 3 4; do
    echo "==== n=$n QSIM_PLAN_FORCE_LOW=$low" >> $out
    QSIM_PLAN_FORCE_LOW=$low python3 tools/step_times.py $n 20260228 1 2 3 >> $out 2>&1
  done
done
echo "==== per-pass times, n=28, FORCE_LOW=4" >> $out
QSIM_PLAN_FORCE_LOW=4 python3 tools/pass_times.py 28 40 20260228 2>&1 | grep -A200 "second execution" >> $out
echo "==== per-pass times, n=28, FORCE_LOW=0" >> $out
python3 tools/pass_times.py 28 40 20260228 2>&1 | grep -A200 "second execution" >> $out
echo done
