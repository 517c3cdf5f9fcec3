#!/bin/bash
# Engine-side planning knobs A/B at a cache-resident size (engine bound) and at 28 qubits (probe build):
#   bash tools/knob_ab.sh "24 28"
R=$(cd "$(dirname "$0")/.." && pwd)
cd $R
for n in ${1:-24 28}; do
  for kn in "" "QSIM_TILE_COMMUTE_FUSE=2" "QSIM_TILE_COMMUTE_FUSE=1" "QSIM_TILE_MUX=0" "QSIM_TILE_SINK_SWAPS=0" "QSIM_TILE_HAD=0" "QSIM_TILE_DIRECT=0" "QSIM_TILE_GROUP_SEARCH=0" "QSIM_TILE_SPECIAL=0" "QSIM_DEBUG_SKIP_GATES=1"; do
    printf "n=%s %-28s " $n "${kn:-default}"
    if [ -n "$kn" ]; then export $kn; fi
    python3 tools/step_times.py $n 20260228 1 2 3 | tail -1
    if [ -n "$kn" ]; then unset ${kn%%=*}; fi
  done
done
