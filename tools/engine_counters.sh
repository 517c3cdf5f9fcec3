#!/bin/bash
# Instruction-side counters of the fused tile pass WITH its gates (bench circuit, probe build through
# tools/tile_pass_once.py): instruction cache, scalar data cache (the record stream), issue / wait split.
#   bash tools/engine_counters.sh <tag> [sizes, default "28 24"]
set -u
tag=$1
sizes=${2:-"28 24"}
R=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp
cd /tmp
OUT=$R/gpurun_out/${tag}_engine_counters
rm -rf $OUT; mkdir -p $OUT
groups=(
 "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE"
 "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_MISSES_DUPLICATE"
 "SQ_IFETCH SQ_IFETCH_LEVEL SQC_ICACHE_BUSY_CYCLES SQC_DCACHE_BUSY_CYCLES"
 "SQC_TC_STALL SQC_TC_INST_REQ SQC_TC_DATA_READ_REQ SQC_TC_REQ"
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES"
 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS"
 "SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM"
 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_LDS_BANK_CONFLICT"
)
for n in $sizes; do
  for s in bench "3,4,5,6,12,13,14,15"; do
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
python3 tools/tile_mem_counters_summary.py $OUT > gpurun_out/${tag}_engine_counters.txt
find $OUT -name "*.csv" -size +200k -delete
cat gpurun_out/${tag}_engine_counters.txt
