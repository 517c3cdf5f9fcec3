#!/bin/bash
# SQ instruction-mix counters of the fused tile kernel on the default bench workload (separate
# rocprofv3 --pmc passes, summary -> gpurun_out/<tag>_k_tile_sq_counters.json):
#   bash tools/sq_counters.sh <tag>
set -e
tag=$1
R=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp
cd /tmp
i=0
for set in "SQ_INSTS_SALU SQ_INSTS_VALU SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_SMEM SQ_INSTS_LDS" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU" "SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM" "SQ_INSTS_VMEM SQ_INSTS_BRANCH" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/sq_$i -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-sweep --fused-qubits 0 --sustain-seconds 0 > /dev/null 2> $R/gpurun_out/sq_$i.err || echo "set '$set' failed"
done
cd $R
python3 - "$tag" <<'PY'
import collections, csv, glob, json, sys
tag = sys.argv[1]
acc = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob("gpurun_out/sq_*/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_tile" in r["Kernel_Name"]:
            a = acc[r["Counter_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
avg = {k: v[1] / v[0] for k, v in sorted(acc.items())}
waves = avg.get("SQ_WAVES", 0) or 1
doc = {"kernel": "k_tile<11>", "source": "rocprofv3 --pmc SQ_* on bench.py (28-qubit workload), averages per launch",
       "counters": avg, "per_wave": {k: v / waves for k, v in avg.items() if k != "SQ_WAVES"}}
json.dump(doc, open(f"gpurun_out/{tag}_k_tile_sq_counters.json", "w"), indent=1)
print(json.dumps(doc["per_wave"], indent=1))
PY
rm -rf gpurun_out/sq_[0-9]*
