#!/bin/bash
# Probe build: the dense-block kernel k_dense_mfma2 under its launch knobs, each setting one run of
# tools/dense_block_probe.py at n qubits, same box back to back:
#   QSIM_DENSE_PF      how the next column group is requested (0 not ahead, 1 if there is one, 2 unconditionally)
#   QSIM_DENSE_GROUPS  column groups per wave, k <= 4;  QSIM_DENSE_CONSEC=1: consecutive ones (else strided through the XCD's region)
#   QSIM_DENSE_WGS     workgroups per CU of the k = 5 grid;  QSIM_DENSE_SKEW: XCD x starts x * skew groups into its region
#   tools/dense_knob_scan.sh [n] [scan]  > profiles/<tag>_dense_knob_scan.txt
n=${1:-30}
scan=${2:-1}
cd "$(dirname "$0")/.."
export QSIM_LIBRARY=$PWD/quantum_simulations_amd/libqsim_hip_probes.so
run() { echo "== $*"; (export "$@"; timeout -k 10 200 python tools/dense_block_probe.py $n) || exit 1; }
if [ "$scan" = 1 ]; then
  for s in "QSIM_DENSE_PF=1" "QSIM_DENSE_PF=2" "QSIM_DENSE_PF=0" "QSIM_DENSE_PF=1 QSIM_DENSE_GROUPS=1" "QSIM_DENSE_PF=1 QSIM_DENSE_GROUPS=2" \
           "QSIM_DENSE_PF=1 QSIM_DENSE_GROUPS=16" "QSIM_DENSE_PF=1 QSIM_DENSE_GROUPS=64" "QSIM_DENSE_PF=0 QSIM_DENSE_GROUPS=1" \
           "QSIM_DENSE_PF=1 QSIM_DENSE_WGS=2" "QSIM_DENSE_PF=1 QSIM_DENSE_WGS=3" "QSIM_DENSE_PF=2 QSIM_DENSE_WGS=2" "QSIM_DENSE_PF=2 QSIM_DENSE_WGS=3" "QSIM_DENSE_PF=1"; do
    run $s
  done
elif [ "$scan" = 2 ]; then       # second scan: k = 5 with fewer resident workgroups; k = 3, 4 by groups per wave, consecutive or strided, XCD streams out of step
  export DENSE_MORE_SETS=1
  for s in "QSIM_DENSE_PF=1" "QSIM_DENSE_PF=1 QSIM_DENSE_WGS=1" "QSIM_DENSE_PF=1 QSIM_DENSE_WGS=2" "QSIM_DENSE_PF=0 QSIM_DENSE_WGS=1" "QSIM_DENSE_PF=0 QSIM_DENSE_WGS=2" "QSIM_DENSE_PF=0 QSIM_DENSE_WGS=3" \
           "QSIM_DENSE_PF=0 QSIM_DENSE_WGS=2 QSIM_DENSE_SKEW=1000003"; do
    DENSE_KS=5 run $s
  done
  for s in "QSIM_DENSE_PF=1" "QSIM_DENSE_PF=1 QSIM_DENSE_GROUPS=2" "QSIM_DENSE_PF=1 QSIM_DENSE_GROUPS=3" "QSIM_DENSE_PF=1 QSIM_DENSE_GROUPS=8" "QSIM_DENSE_PF=0 QSIM_DENSE_GROUPS=2" \
           "QSIM_DENSE_PF=1 QSIM_DENSE_GROUPS=2 QSIM_DENSE_CONSEC=1" "QSIM_DENSE_PF=1 QSIM_DENSE_GROUPS=4 QSIM_DENSE_CONSEC=1" "QSIM_DENSE_PF=1 QSIM_DENSE_GROUPS=8 QSIM_DENSE_CONSEC=1" \
           "QSIM_DENSE_PF=1 QSIM_DENSE_GROUPS=2 QSIM_DENSE_SKEW=100003" "QSIM_DENSE_PF=1 QSIM_DENSE_GROUPS=4 QSIM_DENSE_SKEW=100003" "QSIM_DENSE_PF=1 QSIM_DENSE_GROUPS=2 QSIM_DENSE_SKEW=4099" "QSIM_DENSE_PF=1"; do
    DENSE_KS=3,4 run $s
  done
fi
if [ "$scan" = 3 ]; then   # third scan: runs of consecutive column groups (QSIM_DENSE_CONSEC = run length) for every k
  export DENSE_MORE_SETS=1
  for s in "QSIM_DENSE_PF=0 QSIM_DENSE_WGS=1 QSIM_DENSE_CONSEC=1" "QSIM_DENSE_PF=0 QSIM_DENSE_WGS=1 QSIM_DENSE_CONSEC=2" "QSIM_DENSE_PF=0 QSIM_DENSE_WGS=1 QSIM_DENSE_CONSEC=4" "QSIM_DENSE_PF=0 QSIM_DENSE_WGS=1 QSIM_DENSE_CONSEC=8" \
           "QSIM_DENSE_PF=1 QSIM_DENSE_WGS=2 QSIM_DENSE_CONSEC=1" "QSIM_DENSE_PF=1 QSIM_DENSE_WGS=2 QSIM_DENSE_CONSEC=2" "QSIM_DENSE_PF=1 QSIM_DENSE_WGS=2 QSIM_DENSE_CONSEC=4" "QSIM_DENSE_PF=0 QSIM_DENSE_WGS=2 QSIM_DENSE_CONSEC=4" \
           "QSIM_DENSE_PF=2 QSIM_DENSE_WGS=1 QSIM_DENSE_CONSEC=4" "QSIM_DENSE_PF=1 QSIM_DENSE_WGS=1 QSIM_DENSE_CONSEC=4"; do
    DENSE_KS=5 run $s
  done
  for s in "QSIM_DENSE_PF=0 QSIM_DENSE_CONSEC=1" "QSIM_DENSE_PF=0 QSIM_DENSE_CONSEC=2" "QSIM_DENSE_PF=0 QSIM_DENSE_CONSEC=4" "QSIM_DENSE_PF=2 QSIM_DENSE_CONSEC=4" "QSIM_DENSE_PF=1 QSIM_DENSE_CONSEC=1"; do
    DENSE_KS=6 run $s
  done
  for s in "QSIM_DENSE_PF=1 QSIM_DENSE_CONSEC=1" "QSIM_DENSE_PF=1 QSIM_DENSE_CONSEC=4" "QSIM_DENSE_PF=1 QSIM_DENSE_CONSEC=8 QSIM_DENSE_GROUPS=8" "QSIM_DENSE_PF=1 QSIM_DENSE_CONSEC=4 QSIM_DENSE_GROUPS=8" \
           "QSIM_DENSE_PF=1 QSIM_DENSE_CONSEC=2 QSIM_DENSE_GROUPS=4" "QSIM_DENSE_PF=0 QSIM_DENSE_CONSEC=4" "QSIM_DENSE_PF=1 QSIM_DENSE_CONSEC=4"; do
    DENSE_KS=3,4 run $s
  done
fi
