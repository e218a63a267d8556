#!/bin/bash
# round 5: the one-trajectory tile -- parity tests, then timings against the 4-trajectory tile
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_gpu_round5.py tests/test_gpu_round4.py -x -q > gpurun_out/r5_row1_tests.log 2>&1; rc=$?
tail -15 gpurun_out/r5_row1_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python3 tools/bench_small_tiles.py > gpurun_out/r5_small_tiles.log 2>&1; rc=$?
cat gpurun_out/r5_small_tiles.log | grep -v "^{"
exit $rc
