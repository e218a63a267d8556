cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
{
for C in "--model hh --batch 393216" "--model hh --batch 131072" "--model m6 --batch 65536" "--model nnf --batch 262144" "--model nnf --batch 65536"; do
  echo "== stamps $C"
  IONODE_LIB=$GRAFT_REPO_ROOT/neural-ode-ion-channels_amd/variants/stamps/libionode.so timeout -k 10 200 python3 tools/bench_closed_form.py $C --nt 20001 --reps 1 --stamps 2>&1 | grep -i "STAMPS\|Error" | cut -c1-600
done
} > gpurun_out/r4_stamps2.log 2>&1
cat gpurun_out/r4_stamps2.log
echo "== T4 stamps (4 trajectories, one tile)"; IONODE_LIB=$GRAFT_REPO_ROOT/neural-ode-ion-channels_amd/variants/stamps/libionode.so timeout -k 10 200 python3 bench.py --stamps --batch 4 --nt 20001 --steps 1 --warmup 0 --no-cpu-baseline --no-extra-legs --tile-waves 2 2>&1 | grep STAMPS | cut -c1-900
echo "== T16 stamps (16 trajectories, one tile)"; IONODE_LIB=$GRAFT_REPO_ROOT/neural-ode-ion-channels_amd/variants/stamps/libionode.so timeout -k 10 200 python3 bench.py --stamps --batch 16 --nt 20001 --steps 1 --warmup 0 --no-cpu-baseline --no-extra-legs --tile-waves 4 2>&1 | grep STAMPS | cut -c1-900
