cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
for rep in 1 2; do
for d in neural-ode-ion-channels_amd/variants/*/ neural-ode-ion-channels_amd/; do
  n=$(basename $d)
  r=$(IONODE_LIB=$GRAFT_REPO_ROOT/$d/libionode.so timeout -k 10 300 python3 tools/bench_regression.py 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_iteration'], d['frac_of_fp32_mfma_peak'])")
  echo "$n $r"
done
done
