#!/bin/bash
# Memory-side counters of the fused tile pass by tile-bit set: gate-less passes over a fast
# and a slow set of tile bits, the contiguous tile, the bench passes with their gates and the copy kernel, at an
# HBM-resident (28 qubits) and an Infinity-Cache-resident (24 qubits) state size, each under separate
# `rocprofv3 --pmc` passes (TCP / UTCL1 / TCC counters).  Probe build.  Output: gpurun_out/<tag>_tile_mem_counters.txt
#   bash tools/tile_mem_counters.sh <tag> [sizes, default "28 24"]
set -u
tag=$1
sizes=${2:-"28 24"}
R=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp
cd /tmp
OUT=$R/gpurun_out/${tag}_tile_mem_counters
rm -rf $OUT; mkdir -p $OUT
groups=(
 "TCP_UTCL1_REQUEST TCP_UTCL1_TRANSLATION_HIT TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS"
 "TCP_UTCL1_STALL_INFLIGHT_MAX TCP_UTCL1_STALL_MULTI_MISS TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS TCP_UTCL1_LFIFO_FULL"
 "TCP_PENDING_STALL_CYCLES TCP_TCR_TCP_STALL_CYCLES TCP_GATE_EN1 TCP_GATE_EN2"
 "TCP_TCC_READ_REQ_LATENCY TCP_TCC_READ_REQ TCP_TCC_WRITE_REQ_LATENCY TCP_TCC_WRITE_REQ"
 "TCP_TCP_TA_DATA_STALL_CYCLES TCP_TA_TCP_STATE_READ TCP_TD_TCP_STALL_CYCLES TCP_LFIFO_STALL_CYCLES"
 "TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum TCC_BUBBLE_sum"
 "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_BUSY_sum"
 "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum"
 "TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_SRC_FIFO_FULL_sum TCC_IB_STALL_sum TCC_LATENCY_FIFO_FULL_sum"
 "TCC_REQ_sum TCC_STREAMING_REQ_sum TCC_NC_REQ_sum TCC_CYCLE_sum"
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM"
)
for n in $sizes; do
  hi=$((n-8)); h1=$((n-7)); h2=$((n-6)); h3=$((n-5))
  sets=("copy" "bench" "3,4,5,6,7,8,9,10" "3,4,5,6,12,13,14,15" "3,4,5,6,16,17,18,19" "3,4,5,6,$hi,$h1,$h2,$h3" "3,4,5,6,$((n-4)),$((n-3)),$((n-2)),$((n-1))" "$((n-8)),$((n-7)),$((n-6)),$((n-5)),$((n-4)),$((n-3)),$((n-2)),$((n-1))")
  for s in "${sets[@]}"; do
    name=$(echo "n${n}_$s" | tr ',' '-')
    python3 $R/tools/tile_pass_once.py $n $s 8 > $OUT/$name.time 2>&1
    echo "$name $(tail -1 $OUT/$name.time)"
    gi=0
    for g in "${groups[@]}"; do
      gi=$((gi+1))
      rocprofv3 --pmc $g --kernel-trace --output-format csv -d $OUT/${name}_g$gi -- python3 $R/tools/tile_pass_once.py $n $s 3 > $OUT/${name}_g$gi.log 2>&1 || echo "  group $gi failed for $name"
    done
  done
done
cd $R
python3 tools/tile_mem_counters_summary.py $OUT > gpurun_out/${tag}_tile_mem_counters.txt
find $OUT -name "*.csv" -size +200k -delete
tail -n 80 gpurun_out/${tag}_tile_mem_counters.txt
