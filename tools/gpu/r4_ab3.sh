# round 4: in-tree library vs the arithmetic-cursor and four-per-SIMD 5x10-net builds; T4 phase stamps
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
{
for d in neural-ode-ion-channels_amd/variants/arith/ neural-ode-ion-channels_amd/variants/t64w4/ neural-ode-ion-channels_amd/; do
  n=$(basename $d)
  for a in "--model hh --batch 393216" "--model hh --batch 524288" "--model hh --batch 262144" "--model hh --batch 131072" "--model m6 --batch 65536" "--model m6 --batch 131072" "--model m6 --batch 262144" "--model nnf --batch 65536" "--model nnf --batch 262144" "--model nnf --batch 524288"; do
    if [ $n = t64w4 ] && [[ "$a" != *nnf* ]]; then continue; fi
    IONODE_LIB=$GRAFT_REPO_ROOT/$d/libionode.so timeout -k 10 200 python3 tools/bench_closed_form.py --nt 20001 --reps 3 $a 2>/dev/null | python3 -c "
import sys,json
r=json.load(sys.stdin); print('$n $a', r['kernel'][-28:], round(r['ms'],2), round(r['frac_of_8TBps'],4), r['ok'])"
  done
done
for st in 1 2 4; do
  for a in "--model hh --batch 262144" "--model hh --batch 393216" "--model hh --batch 524288" "--model m6 --batch 131072" "--model nnf --batch 262144"; do
    IONODE_LW_STAGGER=$st timeout -k 10 200 python3 tools/bench_closed_form.py --nt 20001 --reps 3 $a 2>/dev/null | python3 -c "
import sys,json
r=json.load(sys.stdin); print('stagger$st $a', r['kernel'][-28:], round(r['ms'],2), round(r['frac_of_8TBps'],4), r['ok'])"
  done
done
echo "== T4 stamps (4 trajectories, one tile)"; IONODE_LIB=$GRAFT_REPO_ROOT/neural-ode-ion-channels_amd/variants/stamps/libionode.so timeout -k 10 200 python3 bench.py --stamps --batch 4 --nt 20001 --steps 1 --warmup 0 --no-cpu-baseline --no-extra-legs --tile-waves 2 2>&1 | grep STAMPS | cut -c1-900
echo "== T4 stamps (1024 trajectories)"; IONODE_LIB=$GRAFT_REPO_ROOT/neural-ode-ion-channels_amd/variants/stamps/libionode.so timeout -k 10 200 python3 bench.py --stamps --batch 1024 --nt 20001 --steps 1 --warmup 0 --no-cpu-baseline --no-extra-legs --tile-waves 2 2>&1 | grep STAMPS | cut -c1-900
echo "== T16 stamps (16 trajectories, one tile)"; IONODE_LIB=$GRAFT_REPO_ROOT/neural-ode-ion-channels_amd/variants/stamps/libionode.so timeout -k 10 200 python3 bench.py --stamps --batch 16 --nt 20001 --steps 1 --warmup 0 --no-cpu-baseline --no-extra-legs --tile-waves 4 2>&1 | grep STAMPS | cut -c1-900
} > gpurun_out/r4_ab3.log 2>&1
cat gpurun_out/r4_ab3.log
