cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
for G in 6 12 24 48 96; do
  echo "== budget $G GB"
  timeout -k 10 300 python3 tools/bench_grad.py --reps 2 --budget-gb $G 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['forward_with_checkpoints_s'], d['backward_s'], d['grad_w_norm'])"
done
