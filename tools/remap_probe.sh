cd $GRAFT_REPO_ROOT
echo "== default"; python3 tools/step_times.py 28 20260228 1 2 3 | tail -1
for set in "3,4,5,6,12,13,14,15" "3,4,5,6,11,12,13,14" "3,4,5,6,7,8,9,10" "3,4,5,6,20,21,22,23"; do
  echo "== remap $set"; QSIM_DEBUG_REMAP_TILE=$set python3 tools/step_times.py 28 20260228 1 2 3 | tail -1
  echo "== remap $set gate-less"; QSIM_DEBUG_SKIP_GATES=1 QSIM_DEBUG_REMAP_TILE=$set python3 tools/step_times.py 28 20260228 1 2 3 | tail -1
done
echo "== per pass, default"; python3 tools/pass_times.py 28 40 20260228 2>&1 | grep -A40 "second execution" | grep "timed pass"
echo "== per pass, remap P*"; QSIM_DEBUG_REMAP_TILE=3,4,5,6,12,13,14,15 python3 tools/pass_times.py 28 40 20260228 2>&1 | grep -A40 "second execution" | grep "timed pass"
