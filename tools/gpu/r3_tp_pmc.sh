cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3_tp_pmc; rm -rf $O; mkdir -p $O
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_ANY --output-format csv -d $O/a -- python3 tools/bench_grad.py --reps 1 > /dev/null 2> $O/a.err || exit 1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/b -- python3 tools/bench_grad.py --reps 1 > /dev/null 2> $O/b.err || exit 1
python3 tools/pmc_summary.py $O/a > $O/a.json; python3 tools/pmc_summary.py $O/b > $O/b.json
find $O -name "*counter_collection.csv" -delete
cat $O/a.json | head -60
