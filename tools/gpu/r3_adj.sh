cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 500 python3 -m pytest tests/test_gpu_round3.py tests/test_gpu_grad.py tests/test_gpu_dropin.py tests/test_gpu_end_to_end.py -m gpu -x -q > gpurun_out/r3_adj.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3_adj.log
tail -25 gpurun_out/r3_adj.log | cut -c1-300
