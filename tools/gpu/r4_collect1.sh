# round 4 evidence, first half: net parity subset (the row-pair group size changed last), then bench line + kernel traces + headline counters
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 300 python3 -m pytest tests -m gpu -x -q -k "tiny or lane or nnf or net or s03 or architect or weight_sets or image" > gpurun_out/r4_pk_tests.log 2>&1 || { tail -5 gpurun_out/r4_pk_tests.log; exit 1; }
tail -1 gpurun_out/r4_pk_tests.log
R=r04 PART=1 bash tools/collect_profiles.sh
