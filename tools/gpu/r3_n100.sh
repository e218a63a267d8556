cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 800 python3 -m pytest tests -m gpu -x -q -k "architectures or deep_stacks or s11 or s09 or s10 or fuzz or ensemble or population" > gpurun_out/r3_n100.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3_n100.log
tail -3 gpurun_out/r3_n100.log
