#!/bin/bash
# round 5: slab count of the reduce kernel when it runs BESIDE the recompute kernel (gradient through the solve, configs[4] share)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for rep in 1 2; do for s in 0 46 34 23; do echo "grad slabs override $s"; IONODE_GRAD_SLABS=$s timeout -k 10 200 python3 tools/bench_grad.py --reps 3 2>&1 | grep "^{" | cut -c120-300 || exit 1; done; done
