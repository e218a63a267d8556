cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for v in st_asm; do
  echo "== $v"
  IONODE_LIB=$GRAFT_REPO_ROOT/neural-ode-ion-channels_amd/variants/$v/libionode.so timeout -k 10 200 python3 bench.py --steps 1 --warmup 0 --stamps --no-cpu-baseline --no-extra-legs 2>&1 | grep -i "STAMPS\|Error\|error" | cut -c1-1500
done
