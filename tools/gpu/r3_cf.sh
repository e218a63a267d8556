cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_edge.py tests/test_gpu_configs.py -m gpu -x -q > gpurun_out/r3_t1.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3_t1.log
tail -4 gpurun_out/r3_t1.log
timeout -k 10 400 bash tools/gpu/r3_cf_ab.sh
