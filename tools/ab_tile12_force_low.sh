cd /root/repo
P=$PWD/quantum_simulations_amd
for r in 1 2; do
echo "== tile11 probes, force_low 0 (round $r)"; QSIM_LIBRARY=$P/libqsim_hip_probes.so timeout -k 10 200 python tools/step_times.py 28 20260228 1 2 3
echo "== tile12 probes, force_low 0 (round $r)"; QSIM_LIBRARY=$P/libqsim_hip_tile12_probes.so timeout -k 10 200 python tools/step_times.py 28 20260228 1 2 3
echo "== tile12 probes, force_low 1 (round $r)"; QSIM_PLAN_FORCE_LOW=1 QSIM_LIBRARY=$P/libqsim_hip_tile12_probes.so timeout -k 10 200 python tools/step_times.py 28 20260228 1 2 3
done
echo "== tile12 probes, force_low 2"; QSIM_PLAN_FORCE_LOW=2 QSIM_LIBRARY=$P/libqsim_hip_tile12_probes.so timeout -k 10 200 python tools/step_times.py 28 20260228 1 2 3
echo "== tile11 probes, force_low 1"; QSIM_PLAN_FORCE_LOW=1 QSIM_LIBRARY=$P/libqsim_hip_probes.so timeout -k 10 200 python tools/step_times.py 28 20260228 1 2 3
