cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 800 python3 -m pytest tests/test_gpu_round3.py -m gpu -x -q -k "launch_order or six_state" > gpurun_out/r3_lo.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3_lo.log
tail -5 gpurun_out/r3_lo.log
bash tools/gpu/r3_ab_generic.sh
