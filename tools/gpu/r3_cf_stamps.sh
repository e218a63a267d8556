cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
L=$GRAFT_REPO_ROOT/neural-ode-ion-channels_amd/variants/stamps/libionode.so
for C in "--model m6 --batch 65536" "--model m6 --batch 262144" "--model hh --batch 393216" "--model nnf --batch 65536" "--model nnf --batch 262144"; do
  echo "== $C"
  IONODE_LIB=$L timeout -k 10 200 python3 tools/bench_closed_form.py $C --nt 20001 --reps 1 --stamps 2>&1 | grep -i "STAMPS\|kernel\|Error" | cut -c1-700
done
