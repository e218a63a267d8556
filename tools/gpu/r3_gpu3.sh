cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for rep in 1 2; do timeout -k 10 500 bash tools/ab_variants.sh; done 2>&1 | tee gpurun_out/r3_ab3.log
