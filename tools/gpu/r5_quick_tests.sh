#!/bin/bash
# round 5: a subset of the GPU suite (TESTS = pytest arguments)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 ${LIMIT:-900} python3 -m pytest ${TESTS:-tests -m gpu} -x -q > gpurun_out/r5_quick_tests.log 2>&1; rc=$?
tail -${TAILN:-6} gpurun_out/r5_quick_tests.log
exit $rc
