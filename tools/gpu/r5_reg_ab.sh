#!/bin/bash
# round 5: regression step, one against two workgroups per compute unit (same box)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
: > gpurun_out/r5_reg_ab.log
for rep in 1 2; do
echo "== one workgroup per CU (variants/reg_one)" >> gpurun_out/r5_reg_ab.log
IONODE_LIB=$GRAFT_REPO_ROOT/neural-ode-ion-channels_amd/variants/reg_one/libionode.so IONODE_REGRESS_WG_PER_CU=1 timeout -k 10 120 python3 tools/bench_regression.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/r5_reg_ab.log || exit 1
echo "== two workgroups per CU (in-tree)" >> gpurun_out/r5_reg_ab.log
timeout -k 10 120 python3 tools/bench_regression.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/r5_reg_ab.log || exit 1
done
timeout -k 10 300 python3 -m pytest tests/test_regression.py -x -q -m gpu 2>&1 | tail -2 >> gpurun_out/r5_reg_ab.log
cat gpurun_out/r5_reg_ab.log
