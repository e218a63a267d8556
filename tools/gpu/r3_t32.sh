cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 300 python3 -m pytest tests/test_gpu_round3.py -k "asm_stream" -m gpu -x -q > gpurun_out/r3_t32.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3_t32.log
tail -25 gpurun_out/r3_t32.log
