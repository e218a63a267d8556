#!/bin/bash
# round 5: phase stamps of the one-trajectory tile (diagnostic build: variants/row1_stamps)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export IONODE_LIB=$GRAFT_REPO_ROOT/neural-ode-ion-channels_amd/variants/row1_stamps/libionode.so
timeout -k 10 120 python3 tools/bench_small_tiles.py --batches 1 --tiles ${TILES:-16,2} --stamps --reps 1 2>&1 | grep -v "^{\|amdgpu.ids" > gpurun_out/r5_row1_stamps.log || exit 1
cat gpurun_out/r5_row1_stamps.log
