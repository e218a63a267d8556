cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests/test_gpu_grad.py tests/test_regression.py tests/test_gpu_grad_fuzz.py -m gpu -x -q > gpurun_out/r3_tp.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3_tp.log
tail -4 gpurun_out/r3_tp.log
echo "== two-phase"; timeout -k 10 300 python3 tools/bench_grad.py --reps 2 2>&1 | tail -1 | cut -c1-500
echo "== one-phase"; IONODE_GRAD_ONE_PHASE=1 timeout -k 10 300 python3 tools/bench_grad.py --reps 2 2>&1 | tail -1 | cut -c1-500
