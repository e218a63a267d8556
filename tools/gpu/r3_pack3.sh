cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 800 python3 -m pytest tests -m gpu -x -q -k "markov or m6 or M6 or six or fuzz or closed or objective or current or sse or config or population or table" > gpurun_out/r3_pack3.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3_pack3.log
tail -3 gpurun_out/r3_pack3.log
for d in neural-ode-ion-channels_amd/variants/pack6only neural-ode-ion-channels_amd; do
echo "== objective share $d"
IONODE_LIB=$GRAFT_REPO_ROOT/$d/libionode.so timeout -k 10 300 python3 tools/leg_objective.py > gpurun_out/r3_obj.json 2>gpurun_out/r3_obj.err; python3 -c "
import json; d=json.load(open('gpurun_out/r3_obj.json')); print(json.dumps(d)[:1500])"
done
