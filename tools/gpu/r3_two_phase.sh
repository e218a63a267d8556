cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 300 python3 -m pytest tests/test_gpu_round3.py -m gpu -x -q -k "two_phase" > gpurun_out/r3_tp.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3_tp.log
tail -12 gpurun_out/r3_tp.log
for A in "" "--no-weight-grad"; do
echo "== two-phase $A"; timeout -k 10 300 python3 tools/bench_grad.py --reps 2 $A 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['forward_with_checkpoints_s'], d['backward_s'], d['grad_w_norm'], d['grad_p_norm'])"
done
echo "== one-phase"; IONODE_GRAD_ONE_PHASE=1 timeout -k 10 300 python3 tools/bench_grad.py --reps 2 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['forward_with_checkpoints_s'], d['backward_s'], d['grad_w_norm'], d['grad_p_norm'])"
