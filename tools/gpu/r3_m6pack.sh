cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 700 python3 -m pytest tests -m gpu -x -q -k "markov or m6 or M6 or six or fuzz or closed or state6 or config" > gpurun_out/r3_m6pack.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3_m6pack.log
tail -3 gpurun_out/r3_m6pack.log
for C in "--model m6 --batch 65536" "--model m6 --batch 262144" "--model m6 --batch 16384" "--model m6 --batch 65536 --f32" "--model m6 --batch 131072"; do
  echo "== $C"
  timeout -k 10 200 python3 tools/bench_closed_form.py $C --nt 20001 --reps 3 2>&1 | tail -1 | cut -c1-400
done
