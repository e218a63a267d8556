cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests/test_gpu_round3.py -m gpu -x -q -k "two_phase" > gpurun_out/r3_tpt.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3_tpt.log
tail -15 gpurun_out/r3_tpt.log
