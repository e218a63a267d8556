# round 4: the packed-FMA per-lane net: parity tests first, then the A/B script
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q -k "tiny or lane or nnf or net or s03 or architect or weight_sets or image" > gpurun_out/r4_pk_tests.log 2>&1
rc=$?
tail -4 gpurun_out/r4_pk_tests.log | cut -c1-300
[ $rc -ne 0 ] && exit $rc
bash tools/gpu/r4_ab3.sh
