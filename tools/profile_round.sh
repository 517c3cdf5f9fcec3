#!/bin/bash
# Refresh the judged artifacts of a round on the GPU box (run through gpurun):
#   bash tools/profile_round.sh <tag>      e.g. r01f
# Writes gpurun_out/<tag>_bench.json, gpurun_out/prof_{trace,pmc_fetch,pmc_write}/ and then
# profiles/<tag>_{kernel_stats.csv,pmc_summary.json} + gpurun_out/profiles_<tag>/ copies (gpurun
# merges only gpurun_out/ back, so the summaries are duplicated there).
#   bash tools/profile_round.sh <tag> c3   BASELINE config 3 (one launch per gate at 30 qubits, tools/config3_once.py): the
#                                          per-gate kernels' kernel-trace stats + PMC summary -> profiles/<tag>c3_*
set -e
tag=$1
lq=${2:-28}          # local qubits of the profiled workload (30: the shard size of the multi-GPU runs)
R=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp
mkdir -p $R/gpurun_out
if [ "$lq" = c3 ]; then
  cd /tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_trace -- python3 $R/tools/config3_once.py 30 > $R/gpurun_out/${tag}c3_run.log 2> $R/gpurun_out/prof_trace.err
  echo trace rc=$?
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_pmc_fetch -- python3 $R/tools/config3_once.py 30 > /dev/null 2> $R/gpurun_out/prof_pmc_fetch.err
  echo fetch rc=$?
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_pmc_write -- python3 $R/tools/config3_once.py 30 > /dev/null 2> $R/gpurun_out/prof_pmc_write.err
  echo write rc=$?
  cd $R
  python3 tools/pmc_summary.py ${tag}c3 gpurun_out/prof_trace gpurun_out/prof_pmc_fetch gpurun_out/prof_pmc_write 30
  mkdir -p gpurun_out/profiles_${tag}c3
  cp profiles/${tag}c3_kernel_stats.csv profiles/${tag}c3_pmc_summary.json gpurun_out/profiles_${tag}c3/
  rm -rf gpurun_out/prof_trace gpurun_out/prof_pmc_fetch gpurun_out/prof_pmc_write
  cat profiles/${tag}c3_pmc_summary.json | head -60
  exit 0
fi
if [ "$lq" = 28 ]; then     # the headline line; other sizes only need the rocprofv3 passes below
  cd $R && python3 bench.py --steps 20 --warmup 5 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
  tail -c 600 gpurun_out/${tag}_bench.json; echo
fi
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_trace -- python3 $R/bench.py --local-qubits $lq --steps 3 --warmup 1 --no-cpu-baseline --no-sweep --no-api-path --no-plan-check --fused-qubits 0 --sustain-seconds 0 > $R/gpurun_out/${tag}_bench_under_rocprof.json 2> $R/gpurun_out/prof_trace.err
echo trace rc=$?
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_pmc_fetch -- python3 $R/bench.py --local-qubits $lq --steps 1 --warmup 0 --no-cpu-baseline --no-sweep --no-api-path --no-plan-check --fused-qubits 0 --sustain-seconds 0 > /dev/null 2> $R/gpurun_out/prof_pmc_fetch.err
echo fetch rc=$?
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_pmc_write -- python3 $R/bench.py --local-qubits $lq --steps 1 --warmup 0 --no-cpu-baseline --no-sweep --no-api-path --no-plan-check --fused-qubits 0 --sustain-seconds 0 > /dev/null 2> $R/gpurun_out/prof_pmc_write.err
echo write rc=$?
cd $R
python3 tools/pmc_summary.py $tag gpurun_out/prof_trace gpurun_out/prof_pmc_fetch gpurun_out/prof_pmc_write $lq
mkdir -p gpurun_out/profiles_$tag
cp profiles/${tag}_kernel_stats.csv profiles/${tag}_pmc_summary.json gpurun_out/profiles_$tag/
rm -rf gpurun_out/prof_trace gpurun_out/prof_pmc_fetch gpurun_out/prof_pmc_write
cat profiles/${tag}_pmc_summary.json | head -40
