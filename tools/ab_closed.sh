#!/bin/bash
# Dev tool (GPU box): closed-form kernels, in-tree library against variants/*
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
for d in neural-ode-ion-channels_amd/variants/*/ neural-ode-ion-channels_amd/; do
  n=$(basename $d)
  for a in "--batch 196608" "--batch 393216" "--batch 196608 --f32" "--batch 196608 --sse" "--model m6 --batch 65536" "--model m6 --batch 65536 --f32"; do
    IONODE_LIB=$GRAFT_REPO_ROOT/$d/libionode.so python3 tools/bench_closed_form.py --nt 20001 --reps 2 $a 2>/dev/null | python3 -c "
import sys,json
r=json.load(sys.stdin); print('$n $a', r['kernel'][-28:], round(r['ms'],2), r['ok'])"
  done
done
