# round 4: what bounds the 2-state kernel now?  stamps of the current build, instruction / LDS counters of base and current
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4_hh_pmc; mkdir -p $O
{
for C in "--model hh --batch 393216" "--model m6 --batch 65536" "--model nnf --batch 262144"; do
  echo "== stamps $C"
  IONODE_LIB=$GRAFT_REPO_ROOT/neural-ode-ion-channels_amd/variants/stamps/libionode.so timeout -k 10 200 python3 tools/bench_closed_form.py $C --nt 20001 --reps 1 --stamps 2>&1 | grep -i "STAMPS\|Error" | cut -c1-600
done
for v in base ""; do
  L=$GRAFT_REPO_ROOT/neural-ode-ion-channels_amd/libionode.so; [ -n "$v" ] && L=$GRAFT_REPO_ROOT/neural-ode-ion-channels_amd/variants/$v/libionode.so
  export IONODE_LIB=$L
  for C in "--model hh --batch 393216" "--model m6 --batch 65536"; do
    t=$(echo $C | tr -d ' -')
    rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_BUSY_CYCLES --output-format csv -d $O/a_${v}_$t -- python3 tools/bench_closed_form.py $C --nt 20001 --reps 1 > /dev/null 2> $O/a_${v}_$t.err
    rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $O/b_${v}_$t -- python3 tools/bench_closed_form.py $C --nt 20001 --reps 1 > /dev/null 2> $O/b_${v}_$t.err
    echo "== variant [$v] $C"; python3 tools/pmc_summary.py $O/a_${v}_$t | grep -v "^{\|^}" ; python3 tools/pmc_summary.py $O/b_${v}_$t | grep -v "^{\|^}"
  done
done
find $O -name "*counter_collection.csv" -delete
} > gpurun_out/r4_hh_pmc.log 2>&1
tail -80 gpurun_out/r4_hh_pmc.log
