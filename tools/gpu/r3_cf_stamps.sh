cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
L=$GRAFT_REPO_ROOT/neural-ode-ion-channels_amd/variants/stamps/libionode.so
while IFS= read -r C; do
  [ -z "$C" ] && continue
  echo "== $C"
  IONODE_LIB=$L timeout -k 10 200 python3 tools/bench_closed_form.py $C --nt 20001 --reps 1 --stamps 2>&1 | grep -i "STAMPS\|kernel\|Error" | cut -c1-700
done < tools/gpu/cases.txt
