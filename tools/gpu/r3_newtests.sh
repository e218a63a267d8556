cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 1000 python3 -m pytest tests/test_gpu_round3.py tests/test_gpu_grad_fuzz.py -m gpu -q -s --durations=0 > gpurun_out/r3_newtests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3_newtests.log
grep -v "^$" gpurun_out/r3_newtests.log | tail -60
