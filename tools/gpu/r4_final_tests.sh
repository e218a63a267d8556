# round 4: the gradient fuzz log (ADVICE: regenerated from HEAD, summary only) and the new round-4 tests
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_gpu_grad_fuzz.py -q > gpurun_out/r04_grad_fuzz_full.log 2>&1; rc1=$?
{ echo "# python -m pytest tests/test_gpu_grad_fuzz.py -q   (round 4 HEAD: two-phase sweep default, forward on the 4-trajectory tile)"; tail -3 gpurun_out/r04_grad_fuzz_full.log; } > gpurun_out/r04_grad_fuzz.log
cat gpurun_out/r04_grad_fuzz.log
timeout -k 10 600 python3 -m pytest tests/test_gpu_round4.py -x -q 2>&1 | tail -5
exit $rc1
