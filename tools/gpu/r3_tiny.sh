cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q -k "architectures or tiny or fuzz or s03 or population or launch_order or ensemble" > gpurun_out/r3_tiny.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3_tiny.log
tail -3 gpurun_out/r3_tiny.log
