#!/bin/bash
# Matrix-core counters of the dense-block kernels (k_dense_mfma2, k = 3 .. 6) on tools/dense_block_probe.py at 30 qubits: separate
# rocprofv3 --pmc passes (kernel trace only), summary -> gpurun_out/<tag>_dense_mfma_counters.json
#   bash tools/dense_mfma_counters.sh <tag>
set -e
tag=$1
R=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp
cd /tmp
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA" "SQ_INSTS_VALU SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/dm_$i -- python3 $R/tools/dense_block_probe.py 30 > /dev/null 2> $R/gpurun_out/dm_$i.err || echo "set '$set' failed"
done
cd $R
python3 - "$tag" <<'PY'
import collections, csv, glob, json, sys
tag = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob("gpurun_out/dm_*/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if "k_dense" in name:
            a = acc[name.split("(")[0]][r["Counter_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
doc = {"source": "rocprofv3 --pmc (separate passes) on tools/dense_block_probe.py 30, averages per launch",
       "kernels": {k: {c: v[1] / v[0] for c, v in sorted(cs.items())} for k, cs in sorted(acc.items())}}
json.dump(doc, open(f"gpurun_out/{tag}_dense_mfma_counters.json", "w"), indent=1)
print(json.dumps(doc, indent=1))
PY
rm -rf gpurun_out/dm_[0-9]*
