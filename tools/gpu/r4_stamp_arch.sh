cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
{
for lib in variants/stamps/libionode.so libionode.so; do
for a in "--layers 5 --width 100" "--layers 5 --width 100 --out-stride 100" "--layers 5 --width 200 --tile-waves 4" "--layers 5 --width 200 --tile-waves 4 --out-stride 100"; do
  echo "== $lib $a"
  IONODE_LIB=$GRAFT_REPO_ROOT/neural-ode-ion-channels_amd/$lib timeout -k 10 120 python3 tools/stamp_arch.py $a 2>&1 | tail -3
done
done
} > gpurun_out/r4_stamp_arch.log 2>&1
cat gpurun_out/r4_stamp_arch.log
