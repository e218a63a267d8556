# round 4 evidence, second half: the write-bandwidth probe, then the lane-wise kernels' counters + summaries
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
bash tools/gpu/r4_hbm_write.sh || exit 1
R=r04 PART=2 bash tools/collect_profiles.sh
