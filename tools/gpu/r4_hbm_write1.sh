cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
{ echo "== 1 workgroup per CU (163840 B)"; timeout -k 10 120 tools/ubench/hbm_write 393216 20001 163840; } > gpurun_out/r4_hbm_write1.log 2>&1
grep -E "rows6|rows s=34 |==" gpurun_out/r4_hbm_write1.log
