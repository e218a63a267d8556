#!/bin/bash
# Dev tool (GPU box): time the headline kernel with every library under neural-ode-ion-channels_amd/variants/*/ (IONODE_LIB override)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
for d in neural-ode-ion-channels_amd/variants/*/ neural-ode-ion-channels_amd/; do
  n=$(basename $d)
  IONODE_LIB=$GRAFT_REPO_ROOT/$d/libionode.so python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra-legs 2>/dev/null | python3 -c "
import sys,json
r=json.load(sys.stdin); print('$n', round(r['roofline']['kernel_ms'],2), round(r['roofline']['frac'],4), r['config']['trajectories_ok'], r['config']['loss'])" || echo "$n failed"
done
