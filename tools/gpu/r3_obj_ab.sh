cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_edge.py tests/test_gpu_configs.py tests/test_gpu_round3.py -m gpu -x -q > gpurun_out/r3_t1.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3_t1.log
tail -3 gpurun_out/r3_t1.log
for d in neural-ode-ion-channels_amd/variants/*/ neural-ode-ion-channels_amd/; do
  n=$(basename $d)
  IONODE_LIB=$GRAFT_REPO_ROOT/$d/libionode.so timeout -k 10 300 python3 tools/leg_objective.py 2>/dev/null | python3 -c "
import sys,json
r=json.load(sys.stdin)
for k in ('cmaes_first_generation','prior_box'):
    v=r[k]; print('$n', k, {f: round(v[f]['ms'],1) for f in ('pr3','pr4','pr5')}, 'share ms', round(v['ms_per_generation_share'],1))
print('$n nnf', round(r['nnf_s00_candidates_pr5']['ms'],1))"
done
