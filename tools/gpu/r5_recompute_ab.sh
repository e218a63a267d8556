#!/bin/bash
# round 5: gradient share, recompute kernel at one against two workgroups per compute unit (same box)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
: > gpurun_out/r5_recompute_ab.log
for v in "" recompute2 "" recompute2; do
  echo "== variant: ${v:-in-tree}" >> gpurun_out/r5_recompute_ab.log
  if [ -n "$v" ]; then export IONODE_LIB=$GRAFT_REPO_ROOT/neural-ode-ion-channels_amd/variants/$v/libionode.so; else unset IONODE_LIB; fi
  timeout -k 10 200 python3 tools/bench_grad.py --reps 2 2>&1 | grep "^{" | cut -c100-330 >> gpurun_out/r5_recompute_ab.log || exit 1
done
cat gpurun_out/r5_recompute_ab.log
