cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_kats.py -m gpu -x -q > gpurun_out/r3_t1.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3_t1.log
tail -5 gpurun_out/r3_t1.log
timeout -k 10 400 bash tools/ab_variants.sh > gpurun_out/r3_ab1.log 2>&1
cat gpurun_out/r3_ab1.log
