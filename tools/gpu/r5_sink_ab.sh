#!/bin/bash
# round 5: the gradient / regression unit with and without -sink-insts-to-avoid-spills (same box)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
: > gpurun_out/r5_sink_ab.log
for rep in 1 2; do
for v in "" grad_sink; do
  echo "== variant: ${v:-in-tree}" >> gpurun_out/r5_sink_ab.log
  if [ -n "$v" ]; then export IONODE_LIB=$GRAFT_REPO_ROOT/neural-ode-ion-channels_amd/variants/$v/libionode.so; else unset IONODE_LIB; fi
  timeout -k 10 120 python3 tools/bench_regression.py 2>&1 | grep -v amdgpu.ids | cut -c1-200 >> gpurun_out/r5_sink_ab.log || exit 1
  timeout -k 10 200 python3 tools/bench_grad.py --reps 2 2>&1 | grep -v amdgpu.ids | cut -c1-400 >> gpurun_out/r5_sink_ab.log || exit 1
done
done
cat gpurun_out/r5_sink_ab.log
