cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 1000 python3 -m pytest tests/test_gpu_grad.py tests/test_regression.py tests/test_gpu_grad_fuzz.py tests/test_gpu_round3.py tests/test_gpu_end_to_end.py -m gpu -x -q > gpurun_out/r3_tp.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3_tp.log
tail -6 gpurun_out/r3_tp.log
