cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_gpu_grad.py tests/test_regression.py -m gpu -x -q > gpurun_out/r3_t2.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3_t2.log
tail -3 gpurun_out/r3_t2.log
for d in neural-ode-ion-channels_amd/variants/*/ neural-ode-ion-channels_amd/; do
  n=$(basename $d)
  echo "== $n"
  IONODE_LIB=$GRAFT_REPO_ROOT/$d/libionode.so timeout -k 10 300 python3 tools/bench_grad.py --reps 2 2>/dev/null | tail -1 | cut -c1-600
  IONODE_LIB=$GRAFT_REPO_ROOT/$d/libionode.so timeout -k 10 300 python3 tools/bench_regression.py 2>/dev/null | tail -1 | cut -c1-400
done
