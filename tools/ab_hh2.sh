#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
for d in neural-ode-ion-channels_amd/variants/*/ neural-ode-ion-channels_amd/; do
  n=$(basename $d)
  for b in 57344 73728 131072; do for t in 16 64; do
    IONODE_LIB=$GRAFT_REPO_ROOT/$d/libionode.so python3 tools/bench_closed_form.py --nt 100001 --reps 1 --batch $b --sse --f32 --prot 9 --tpw $t 2>/dev/null | python3 -c "
import sys,json
r=json.load(sys.stdin); print('$n B=$b tpw=$t', r['kernel'][-28:], round(r['ms'],2), r['ok'])"
  done; done
done
