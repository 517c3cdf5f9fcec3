set -e
for n in 20 22 24 25; do
  for mb in 0 268435456; do
    for mode in fused per-gate; do
      echo "== n=$n mall=$mb mode=$mode"
      QSIM_MALL_BYTES=$mb python bench.py --local-qubits $n --mode $mode --no-sweep --no-cpu-baseline --steps 20 --warmup 3 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('hbm_frac_moved'))"
    done
  done
done
