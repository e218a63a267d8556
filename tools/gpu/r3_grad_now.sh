cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
for A in "" "--no-weight-grad"; do
  echo "== two-phase $A"; timeout -k 10 300 python3 tools/bench_grad.py --reps 2 $A 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['forward_with_checkpoints_s'], d['backward_s'])"
  echo "== one-phase $A"; IONODE_GRAD_ONE_PHASE=1 timeout -k 10 300 python3 tools/bench_grad.py --reps 2 $A 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['forward_with_checkpoints_s'], d['backward_s'])"
done
