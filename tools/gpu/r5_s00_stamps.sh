#!/bin/bash
# round 5: phase stamps of the headline kernel (diagnostic build: variants/s00_stamps)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export IONODE_LIB=$GRAFT_REPO_ROOT/neural-ode-ion-channels_amd/variants/s00_stamps/libionode.so
timeout -k 10 200 python3 bench.py --stamps --steps 1 --warmup 0 --no-cpu-baseline --no-extra-legs > gpurun_out/r5_s00_stamps.json 2> gpurun_out/r5_s00_stamps.log || exit 1
grep STAMPS gpurun_out/r5_s00_stamps.log
