#!/bin/bash
# A/B several builds of libqsim_hip.so in ONE process sequence on ONE device (guide rule 24):
#   tools/ab_libs.sh <rounds> libA.so libB.so ...   (libs inside quantum_simulations_amd/)
rounds=$1; shift
cd "$(dirname "$0")/.."
cp quantum_simulations_amd/libqsim_hip.so /tmp/_orig_lib.so
for r in $(seq 1 $rounds); do
  for lib in "$@"; do
    cp quantum_simulations_amd/$lib quantum_simulations_amd/libqsim_hip.so
    printf "%s round %s: " "$lib" "$r"
    timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-sweep --no-api-path --no-plan-check --fused-qubits ${FUSED_QUBITS:-0} --sustain-seconds 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); f=d.get('fused30') or {}; print(d['value'], d['ms_per_step'], d['config']['hbm_passes_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], '| 30q:', f.get('gate_apps_per_s'), f.get('hbm_passes_per_step'), f.get('avg_launch_ms'), f.get('frac'))"
  done
done
cp /tmp/_orig_lib.so quantum_simulations_amd/libqsim_hip.so
