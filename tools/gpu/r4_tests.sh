# round 4: the whole -m gpu suite (one process), log under gpurun_out/
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q ${PYTEST_ARGS} > gpurun_out/r4_tests.log 2>&1
rc=$?
tail -15 gpurun_out/r4_tests.log
exit $rc
