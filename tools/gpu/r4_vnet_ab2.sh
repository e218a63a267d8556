cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
{
for d in neural-ode-ion-channels_amd/variants/*/ neural-ode-ion-channels_amd/; do
  n=$(basename $d)
  for a in "--model nnf --batch 65536" "--model nnf --batch 131072" "--model nnf --batch 262144" "--model nnf --batch 262144 --f32"; do
    IONODE_LIB=$GRAFT_REPO_ROOT/$d/libionode.so timeout -k 10 200 python3 tools/bench_closed_form.py --nt 20001 --reps 3 $a 2>/dev/null | python3 -c "
import sys,json
r=json.load(sys.stdin); print('$n $a', r['kernel'][-28:], round(r['ms'],2), round(r['frac_of_8TBps'],4), r['ok'])"
  done
done
} > gpurun_out/r4_vnet_ab2.log 2>&1
cat gpurun_out/r4_vnet_ab2.log
