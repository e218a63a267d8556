cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for d in neural-ode-ion-channels_amd/variants/pack6only neural-ode-ion-channels_amd; do
  echo "=== $d"
  for C in "--model m6 --batch 65536" "--model m6 --batch 262144" "--model m6 --batch 131072" "--model m6 --batch 65536 --sse" "--model m6 --batch 262144 --sse" "--model hh --batch 196608 --sse --f32" "--model hh --batch 196608 --current" "--model hh --batch 393216"; do
    echo "== $C"
    IONODE_LIB=$GRAFT_REPO_ROOT/$d/libionode.so timeout -k 10 200 python3 tools/bench_closed_form.py $C --nt 20001 --reps 3 2>&1 | tail -1 | cut -c1-200
  done
done
