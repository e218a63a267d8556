#!/bin/bash
# round 5: the recompute kernel (phase A of the two-phase sweep) at one and two workgroups per compute unit, after the stage inputs moved
# ahead of the product loop.  As run: in-tree = the build of its commit (first run: one per unit; second run: the shipped two-per-unit build
# without the prefetches), variants/recompute2 = tools/ab_one_unit.sh recompute2 ionode_grad_capi.o "-DIONODE_RECOMPUTE_WG_PER_CU=2" of the
# commit before (two per unit WITH the checkpoint / output-gradient prefetches); gradient tests on the in-tree build first
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_gpu_grad.py tests/test_gpu_grad_fuzz.py -x -q -m gpu > gpurun_out/r5_recompute_tests.log 2>&1; rc=$?
tail -3 gpurun_out/r5_recompute_tests.log
[ $rc -eq 0 ] || exit $rc
for rep in 1 2; do
for v in "" recompute2; do
  if [ -n "$v" ]; then export IONODE_LIB=$GRAFT_REPO_ROOT/neural-ode-ion-channels_amd/variants/$v/libionode.so; else unset IONODE_LIB; fi
  echo "== ${v:-in-tree}"
  timeout -k 10 200 python3 tools/bench_grad.py --reps 3 2>&1 | grep "^{" | cut -c120-330 || exit 1
done
done
