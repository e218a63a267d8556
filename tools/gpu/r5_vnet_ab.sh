#!/bin/bash
# round 5: the 5 x 10 per-lane net with a scheduling barrier in the middle of a layer's k loop (fewer scalar registers in flight)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
: > gpurun_out/r5_vnet_ab.log
for rep in 1 2; do
for v in "" $VARIANTS; do
  if [ -n "$v" ]; then export IONODE_LIB=$GRAFT_REPO_ROOT/neural-ode-ion-channels_amd/variants/$v/libionode.so; else unset IONODE_LIB; fi
  for C in "--batch 262144" "--batch 65536" "--batch 262144 --f32"; do
    echo "== ${v:-in-tree} $C" >> gpurun_out/r5_vnet_ab.log
    timeout -k 10 120 python3 tools/bench_closed_form.py --model nnf --nt 20001 --reps 3 $C 2>&1 | grep "^{" | cut -c1-230 >> gpurun_out/r5_vnet_ab.log || exit 1
  done
done
done
cat gpurun_out/r5_vnet_ab.log
