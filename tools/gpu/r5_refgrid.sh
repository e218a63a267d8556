#!/bin/bash
# round 5: one call on the reference's own output grid (linspace(0, 10000, 100001) in fp32: the GENERAL variants)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 200 python3 tools/bench_small_tiles.py --batches 1,32 --tiles 16,2 --ref-grid 2>&1 | grep -v "^{\|amdgpu.ids" > gpurun_out/r5_refgrid.log || exit 1
cat gpurun_out/r5_refgrid.log
