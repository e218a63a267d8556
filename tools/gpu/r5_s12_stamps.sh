#!/bin/bash
# round 5: per-step stamps of the asm stream with and without the short form of k-tile 12
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
: > gpurun_out/r5_s12_stamps.log
for v in s12off_stamps s12_stamps; do
  echo "== $v" >> gpurun_out/r5_s12_stamps.log
  IONODE_LIB=$GRAFT_REPO_ROOT/neural-ode-ion-channels_amd/variants/$v/libionode.so timeout -k 10 200 python3 bench.py --stamps --steps 1 --warmup 0 --no-cpu-baseline --no-extra-legs 2>&1 >/dev/null | grep STAMPS >> gpurun_out/r5_s12_stamps.log || exit 1
done
cat gpurun_out/r5_s12_stamps.log
