# same-box A/B of every library under variants/ against the in-tree one over the closed-form bench cases in $CASES (newline separated)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
while IFS= read -r C; do
  [ -z "$C" ] && continue
  for d in neural-ode-ion-channels_amd/variants/*/ neural-ode-ion-channels_amd/; do
    r=$(IONODE_LIB=$GRAFT_REPO_ROOT/$d/libionode.so timeout -k 10 200 python3 tools/bench_closed_form.py $C --nt 20001 --reps 3 2>&1 | tail -1 | sed -E 's/.*"kernel": "ionode_dopri5_kernel(<[^>]*>).*"ms": ([0-9.]+).*/\1 \2/')
    echo "$C | $(basename $d) | $r"
  done
done < tools/gpu/cases.txt
