#!/bin/bash
# Dev tool (GPU box): in-tree library against variants/base over the MLP architectures (kernel ms)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
for lib in neural-ode-ion-channels_amd/variants/base/libionode.so neural-ode-ion-channels_amd/libionode.so; do
  echo "== $lib"
  IONODE_LIB=$GRAFT_REPO_ROOT/$lib python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra-legs 2>/dev/null | python3 -c "
import sys,json
r=json.load(sys.stdin); print('s00 f64 4096x100001', round(r['roofline']['kernel_ms'],2), round(r['roofline']['frac'],4))"
  for a in "--width 200 --layers 5 --f32" "--width 100 --layers 5" "--width 100 --layers 10 --f32" "--width 500 --layers 5" "--width 500 --layers 1 --f32" "--width 10 --layers 5 --batch 65536" "--width 200 --layers 10"; do
    IONODE_LIB=$GRAFT_REPO_ROOT/$lib python3 tools/bench_closed_form.py --model nnf --batch 4096 --nt 20001 --reps 2 $a 2>/dev/null | python3 -c "
import sys,json
r=json.load(sys.stdin); print('$a', r['kernel'][-22:], round(r['ms'],2), r['ok'])"
  done
done
