cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/r3_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3_tests.log
tail -6 gpurun_out/r3_tests.log
