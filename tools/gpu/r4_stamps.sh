# round 4: phase stamps of the lane-wise kernels (variants/stamps = -DIONODE_STAMPS) + same-box baseline timings of the in-tree library
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
L=$GRAFT_REPO_ROOT/neural-ode-ion-channels_amd/variants/stamps/libionode.so
{
for C in "--model hh --batch 393216" "--model hh --batch 393216 --f32" "--model m6 --batch 65536" "--model nnf --batch 65536" "--model nnf --batch 262144" "--model nnf --batch 262144 --f32" "--model hh --batch 196608 --sse --f32"; do
  echo "== stamps $C"
  IONODE_LIB=$L timeout -k 10 200 python3 tools/bench_closed_form.py $C --nt 20001 --reps 1 --stamps 2>&1 | grep -i "STAMPS\|kernel\|Error" | cut -c1-900
  echo "== base $C"
  timeout -k 10 200 python3 tools/bench_closed_form.py $C --nt 20001 --reps 3 2>&1 | grep -i "kernel\|Error" | cut -c1-700
done
} > gpurun_out/r4_stamps.log 2>&1
tail -40 gpurun_out/r4_stamps.log
