#!/bin/bash
# round 5: the reduce kernel (both light jobs in one pipelined workgroup per slab; N = 200: two column blocks, two workgroups per compute unit) -- gradient / regression tests, then timings
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 800 python3 -m pytest tests/test_regression.py tests/test_gpu_grad.py tests/test_gpu_grad_fuzz.py tests/test_gpu_end_to_end.py -x -q -m gpu > gpurun_out/r5_reduce_tests.log 2>&1; rc=$?
tail -3 gpurun_out/r5_reduce_tests.log
[ $rc -eq 0 ] || exit $rc
for s in 0; do echo "slabs override $s"; IONODE_REGRESS_SLABS=$s python3 tools/bench_regression.py 2>&1 | grep "^{" | cut -c60-135; done
python3 tools/bench_grad.py --reps 3 2>&1 | grep "^{" | cut -c120-330
