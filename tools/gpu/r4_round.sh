# round 4: stamps + counters of the current build, A/B list, then the GPU suite
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4_hh_pmc; mkdir -p $O
{
for C in "--model hh --batch 393216" "--model m6 --batch 65536" "--model m6 --batch 262144" "--model nnf --batch 262144"; do
  echo "== stamps $C"
  IONODE_LIB=$GRAFT_REPO_ROOT/neural-ode-ion-channels_amd/variants/stamps/libionode.so timeout -k 10 200 python3 tools/bench_closed_form.py $C --nt 20001 --reps 1 --stamps 2>&1 | grep -i "STAMPS\|Error" | cut -c1-600
done
for C in "--model hh --batch 393216" "--model m6 --batch 262144" "--model nnf --batch 262144"; do
  t=$(echo $C | tr -d ' -')
  rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_BUSY_CYCLES --output-format csv -d $O/a_$t -- python3 tools/bench_closed_form.py $C --nt 20001 --reps 1 > /dev/null 2> $O/a_$t.err
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE SQ_INSTS_SMEM SQ_WAVES --output-format csv -d $O/b_$t -- python3 tools/bench_closed_form.py $C --nt 20001 --reps 1 > /dev/null 2> $O/b_$t.err
  echo "== counters $C"; python3 tools/pmc_summary.py $O/a_$t | grep -v "^{\|^}" ; python3 tools/pmc_summary.py $O/b_$t | grep -v "^{\|^}"
done
find $O -name "*counter_collection.csv" -delete
} > gpurun_out/r4_round.log 2>&1
cat gpurun_out/r4_round.log | cut -c1-300
