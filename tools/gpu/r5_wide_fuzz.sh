#!/bin/bash
# round 5: one-off wider seed ranges of the two fuzz sweeps on the final library (forward parity: seeds 0..199; gradients: seeds 100..159)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
IONODE_FUZZ_SEEDS=200 timeout -k 10 500 python3 -m pytest tests/test_gpu_fuzz.py -q -m gpu -x > gpurun_out/r5_wide_fuzz_fwd.log 2>&1; rc=$?
tail -2 gpurun_out/r5_wide_fuzz_fwd.log
[ $rc -eq 0 ] || exit $rc
for s0 in 100 120 140; do
  IONODE_GRAD_FUZZ_SEED0=$s0 timeout -k 10 400 python3 -m pytest tests/test_gpu_grad_fuzz.py -q -m gpu -x > gpurun_out/r5_wide_fuzz_grad_$s0.log 2>&1; rc=$?
  tail -2 gpurun_out/r5_wide_fuzz_grad_$s0.log
  [ $rc -eq 0 ] || exit $rc
done
