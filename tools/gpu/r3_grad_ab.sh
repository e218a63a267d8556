cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for A in "" "--no-weight-grad" "--f64" "--model d2"; do
  echo "== $A"
  timeout -k 10 300 python3 tools/bench_grad.py --reps 2 $A 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['forward_with_checkpoints_s'], d['backward_s'], d['grad_w_norm'])"
done
timeout -k 10 300 python3 tools/bench_regression.py 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_iteration'], d['loss'])"
