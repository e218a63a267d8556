cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_gpu_round3.py -m gpu -x -q -k "rccl" > gpurun_out/r3_rccl.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3_rccl.log
tail -30 gpurun_out/r3_rccl.log
