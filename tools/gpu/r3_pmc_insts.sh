cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3_pmc_insts; mkdir -p $O
S="--steps 1 --warmup 0 --no-cpu-baseline --no-extra-legs"
for v in "" noasm; do
  L=$GRAFT_REPO_ROOT/neural-ode-ion-channels_amd/libionode.so; [ -n "$v" ] && L=$GRAFT_REPO_ROOT/neural-ode-ion-channels_amd/variants/$v/libionode.so
  export IONODE_LIB=$L
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVE_CYCLES --output-format csv -d $O/pmc_$v -- python3 bench.py $S > /dev/null 2> $O/pmc_$v.err
  echo "== variant [$v]"; python3 tools/pmc_summary.py $O/pmc_$v | grep -v "^{\|^}\|dispatches" | head -12
done
