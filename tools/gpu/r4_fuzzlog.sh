cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_gpu_grad_fuzz.py -q > gpurun_out/r04_grad_fuzz_full.log 2>&1; rc=$?
{ echo "# python -m pytest tests/test_gpu_grad_fuzz.py -q   (round 4 final library: two-phase sweep default, forward on the 4-trajectory tile)"; python3 -c "import importlib,sys; sys.path.insert(0,'.'); print('# libionode.so sha256', importlib.import_module('neural-ode-ion-channels_amd').capi.library_digest()[:16])"; tail -3 gpurun_out/r04_grad_fuzz_full.log; } > gpurun_out/r04_grad_fuzz.log
cat gpurun_out/r04_grad_fuzz.log
exit $rc
