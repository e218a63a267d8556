#!/bin/bash
# round 5: the GPU suite, then the default bench line (what the driver runs at round end)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/r5_gputests.log 2>&1; rc=$?
tail -5 gpurun_out/r5_gputests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 280 python3 bench.py > gpurun_out/r5_bench.json 2> gpurun_out/r5_bench.err; rc=$?
head -c 600 gpurun_out/r5_bench.json; echo; tail -3 gpurun_out/r5_bench.err
exit $rc
