cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
{
echo "== mfma 4x4x1 probe"; ./tools/ubench/mfma4x4
for v in nolean ""; do
  L=$GRAFT_REPO_ROOT/neural-ode-ion-channels_amd/libionode.so; [ -n "$v" ] && L=$GRAFT_REPO_ROOT/neural-ode-ion-channels_amd/variants/$v/libionode.so
  echo "== headline [$v]"; IONODE_LIB=$L python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra-legs 2>/dev/null | python3 -c "
import sys,json
r=json.load(sys.stdin); print(r['config']['kernel'], round(r['roofline']['kernel_ms'],2), round(r['roofline']['frac'],4), round(r['value'],1))"
  echo "== 16384 [$v]"; IONODE_LIB=$L python3 bench.py --batch 16384 --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs 2>/dev/null | python3 -c "
import sys,json
r=json.load(sys.stdin); print(r['config']['kernel'], round(r['roofline']['kernel_ms'],2), round(r['roofline']['frac'],4), round(r['value'],1))"
done
} > gpurun_out/r4_step3.log 2>&1
cat gpurun_out/r4_step3.log
