cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 800 python3 -m pytest tests -m gpu -x -q -k "markov or m6 or M6 or six or fuzz or closed or objective or current or sse or config or population or table" > gpurun_out/r3_pack2.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3_pack2.log
tail -3 gpurun_out/r3_pack2.log
for C in "--model m6 --batch 65536" "--model m6 --batch 262144" "--model m6 --batch 65536 --current" "--model hh --batch 196608 --current" "--model hh --batch 196608 --sse --f32" "--model hh --batch 196608 --sse" "--model m6 --batch 65536 --sse" "--model nnf --batch 262144 --current"; do
  echo "== $C"
  timeout -k 10 200 python3 tools/bench_closed_form.py $C --nt 20001 --reps 3 2>&1 | tail -1 | cut -c1-330
done
echo "== objective share"
timeout -k 10 300 python3 tools/leg_objective.py 2>&1 | tail -2 | cut -c1-600
