#!/bin/bash
# Dev tool (GPU box): gradient sweep and regression step with every library under variants/*/
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
for d in neural-ode-ion-channels_amd/variants/*/ neural-ode-ion-channels_amd/; do
  n=$(basename $d)
  IONODE_LIB=$GRAFT_REPO_ROOT/$d/libionode.so python3 tools/bench_grad.py --batch 1024 --nt 100001 --reps 2 2>/dev/null | python3 -c "
import sys,json
r=json.load(sys.stdin); print('$n grad fwd', round(r['forward_with_checkpoints_s'],4), 'bwd', round(r['backward_s'],4), r['grad_w_norm'])" || echo "$n grad failed"
  IONODE_LIB=$GRAFT_REPO_ROOT/$d/libionode.so python3 tools/bench_regression.py 2>/dev/null | python3 -c "
import sys,json
r=json.load(sys.stdin); print('$n regress', json.dumps(r)[:200])" || echo "$n regress failed"
done
