#!/bin/bash
# Shader / memory clocks and socket power while the bench's fused passes run (sampled with rocm-smi from a second
# process): tells whether the pass runs at the 2.4 GHz the cycle models assume.   tools/clock_probe.sh [out]
out=${1:-gpurun_out/clock_probe.txt}
cd "$(dirname "$0")/.."
mkdir -p "$(dirname "$out")"
{ echo "== idle"; rocm-smi --showclocks --showpower 2>&1 | grep -E "sclk|mclk|fclk|Power" ; } > "$out"
timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-sweep --sustain-seconds 12 > gpurun_out/clock_probe_bench.json 2>/dev/null &
bpid=$!
sleep 1
for i in $(seq 1 60); do
  if ! kill -0 $bpid 2>/dev/null; then break; fi
  { echo "== t=$i"; rocm-smi --showclocks --showpower --showuse 2>&1 | grep -E "sclk|mclk|fclk|Power|busy" ; } >> "$out"
  sleep 0.5
done
wait $bpid
echo "bench rc=$?" >> "$out"
