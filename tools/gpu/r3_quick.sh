# quick check: MLP parity tests + A/B of the headline against every variant under neural-ode-ion-channels_amd/variants
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 400 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_kats.py tests/test_gpu_fuzz.py tests/test_gpu_round2.py tests/test_gpu_edge.py tests/test_gpu_configs.py -m gpu -x -q > gpurun_out/r3_t1.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3_t1.log
tail -3 gpurun_out/r3_t1.log
timeout -k 10 400 bash tools/ab_variants.sh 2>&1 | tee gpurun_out/r3_ab.log
