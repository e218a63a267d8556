# re-collect the cf1 passes only (HH, index order) into the existing collection directory
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03_prof; mkdir -p $O
A="--model hh --batch 393216 --index-order --nt 20001 --reps 1"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR --output-format csv -d $O/cf1a -- python3 tools/bench_closed_form.py $A > /dev/null 2> $O/cf1a.err || exit 1
rocprofv3 --pmc GRBM_GUI_ACTIVE FETCH_SIZE --output-format csv -d $O/cf1b -- python3 tools/bench_closed_form.py $A > /dev/null 2> $O/cf1b.err || exit 1
rocprofv3 --pmc WRITE_SIZE TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d $O/cf1c -- python3 tools/bench_closed_form.py $A > /dev/null 2> $O/cf1c.err || exit 1
for p in cf1a cf1b cf1c; do python3 tools/pmc_summary.py $O/$p > $O/$p.json; done
find $O -name "*counter_collection.csv" -delete
cat $O/cf1b.json
