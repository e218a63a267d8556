cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 900 python3 tools/gpu/r4_diag_c3.py > gpurun_out/r4_diag_c3.log 2>&1
tail -40 gpurun_out/r4_diag_c3.log | cut -c1-400
