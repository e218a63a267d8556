# round 4: full -m gpu suite, then row-pair group sizes of the per-lane net (1 / 3 / 5 pairs per group vs the in-tree 2)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/r4_tests.log 2>&1
rc=$?
tail -3 gpurun_out/r4_tests.log | cut -c1-300
[ $rc -ne 0 ] && exit $rc
{
for d in neural-ode-ion-channels_amd/variants/pairs1/ neural-ode-ion-channels_amd/variants/pairs3/ neural-ode-ion-channels_amd/variants/pairs5/ neural-ode-ion-channels_amd/variants/t64w4/ neural-ode-ion-channels_amd/; do
  n=$(basename $d)
  for a in "--model nnf --batch 65536" "--model nnf --batch 262144" "--model nnf --batch 393216" "--model nnf --batch 262144 --f32" "--model m6 --batch 262144" "--model m6 --batch 65536"; do
    if [ $n != neural-ode-ion-channels_amd ] && [[ "$a" != *nnf* ]]; then continue; fi
    IONODE_LIB=$GRAFT_REPO_ROOT/$d/libionode.so timeout -k 10 200 python3 tools/bench_closed_form.py --nt 20001 --reps 3 $a 2>/dev/null | python3 -c "
import sys,json
r=json.load(sys.stdin); print('$n $a', r['kernel'][-28:], round(r['ms'],2), round(r['frac_of_8TBps'],4), r['ok'])"
  done
done
} > gpurun_out/r4_pairs_ab.log 2>&1
cat gpurun_out/r4_pairs_ab.log
