# round 4: 4-trajectory tile tuning: parity subset, phase stamps, single-call latency, configs[4] forward + backward
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_gpu_round4.py tests/test_gpu_round2.py::test_config1_single_sinewave_trajectory_through_the_shim -x -q -s > gpurun_out/r4_t4.log 2>&1
rc=$?
tail -3 gpurun_out/r4_t4.log | cut -c1-300
cat gpurun_out/config1_latency.json; echo
[ $rc -ne 0 ] && exit $rc
echo "== T4 stamps (4 trajectories, one tile)"; IONODE_LIB=$GRAFT_REPO_ROOT/neural-ode-ion-channels_amd/variants/stamps/libionode.so timeout -k 10 200 python3 bench.py --stamps --batch 4 --nt 20001 --steps 1 --warmup 0 --no-cpu-baseline --no-extra-legs --tile-waves 2 2>&1 | grep STAMPS | cut -c1-900
timeout -k 10 300 python3 tools/bench_grad.py --reps 1 --budget-gb 64 > gpurun_out/r4_grad.json 2> gpurun_out/r4_grad.err; tail -1 gpurun_out/r4_grad.json | cut -c1-600
