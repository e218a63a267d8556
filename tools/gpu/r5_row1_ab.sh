#!/bin/bash
# round 5: one-trajectory tile, same-box A/B of build variants (IONODE_LIB selects the library); parity of the in-tree build first
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 300 python3 -m pytest tests/test_gpu_round5.py -x -q -k "one_trajectory or single_odeint" > gpurun_out/r5_row1_tests.log 2>&1; rc=$?
tail -3 gpurun_out/r5_row1_tests.log
[ $rc -eq 0 ] || exit $rc
: > gpurun_out/r5_row1_ab.log
for v in "" $VARIANTS; do
  echo "== variant: ${v:-in-tree}" >> gpurun_out/r5_row1_ab.log
  if [ -n "$v" ]; then export IONODE_LIB=$GRAFT_REPO_ROOT/neural-ode-ion-channels_amd/variants/$v/libionode.so; else unset IONODE_LIB; fi
  timeout -k 10 120 python3 tools/bench_small_tiles.py --batches ${BATCHES:-1} --tiles ${TILES:-16} 2>&1 | grep -v "^{\|amdgpu.ids" >> gpurun_out/r5_row1_ab.log || exit 1
done
cat gpurun_out/r5_row1_ab.log
