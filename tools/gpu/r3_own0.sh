cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 1000 python3 -m pytest tests/test_gpu_grad.py tests/test_regression.py tests/test_gpu_round3.py -m gpu -x -q -k "not fuzz" > gpurun_out/r3_own0.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3_own0.log
tail -4 gpurun_out/r3_own0.log
bash tools/gpu/r3_grad_ab.sh
