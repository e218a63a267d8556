#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
for d in neural-ode-ion-channels_amd/variants/*/ neural-ode-ion-channels_amd/; do
 n=$(basename $d)
 for a in "--batch 65536" "--batch 131072" "--batch 196608" "--batch 262144" "--batch 262144 --layers 10" "--batch 262144 --layers 1"; do
  IONODE_LIB=$GRAFT_REPO_ROOT/$d/libionode.so python3 tools/bench_closed_form.py --model nnf --width 10 --layers 5 --nt 20001 --reps 2 $a --tpw 64 2>/dev/null | python3 -c "
import sys,json
r=json.load(sys.stdin); print('$n $a', r['kernel'][-26:], round(r['ms'],2), round(r['traj_per_s']/1e6,3), round(r['frac_of_8TBps'],4), r['ok'])"
 done
done
