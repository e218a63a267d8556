#!/bin/bash
# round 5: the 5 x 10 per-lane net -- Linear(2, N) / Linear(N, 1) scalars loaded per evaluation + half a layer's weights in flight (in-tree)
# against the round-4 form (variants/vnet_old: -DIONODE_VNET_RELOAD=0 -DIONODE_VNET_SPLIT=0); parity tests of the lane-wise nets first
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q -k "tiny or lane or s03 or s04 or s05 or architectures or small_net or wide or n10 or N10 or nnd" > gpurun_out/r5_vnet_tests.log 2>&1; rc=$?
tail -3 gpurun_out/r5_vnet_tests.log
[ $rc -eq 0 ] || exit $rc
: > gpurun_out/r5_vnet_reload.log
for rep in 1 2; do
for v in "" vnet_old; do
  if [ -n "$v" ]; then export IONODE_LIB=$GRAFT_REPO_ROOT/neural-ode-ion-channels_amd/variants/$v/libionode.so; else unset IONODE_LIB; fi
  for C in "--batch 262144" "--batch 65536" "--batch 262144 --f32" "--batch 393216"; do
    echo "== ${v:-in-tree} $C" >> gpurun_out/r5_vnet_reload.log
    timeout -k 10 120 python3 tools/bench_closed_form.py --model nnf --nt 20001 --reps 3 $C 2>&1 | grep "^{" | cut -c1-230 >> gpurun_out/r5_vnet_reload.log || exit 1
  done
done
done
cut -c1-140 gpurun_out/r5_vnet_reload.log
