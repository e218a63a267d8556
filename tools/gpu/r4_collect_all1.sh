# round 4, final: the -m gpu suite, then evidence part 1 (bench line, kernel traces, headline counters)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/r4_tests.log 2>&1 || { tail -15 gpurun_out/r4_tests.log; exit 1; }
tail -1 gpurun_out/r4_tests.log
R=r04 PART=1 bash tools/collect_profiles.sh
