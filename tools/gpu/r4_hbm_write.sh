cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
{ echo "== unlimited residency"; timeout -k 10 120 tools/ubench/hbm_write 393216 20001 0; echo "== 4 workgroups per CU (34816 B)"; timeout -k 10 120 tools/ubench/hbm_write 393216 20001 34816; echo "== 2 workgroups per CU (75776 B)"; timeout -k 10 120 tools/ubench/hbm_write 393216 20001 75776; } > gpurun_out/r4_hbm_write.log 2>&1
cat gpurun_out/r4_hbm_write.log
