# re-collect the cf5 passes only (HH, default launch order) into the existing collection directory, and run the launch-order tests
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03_prof; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_round3.py -m gpu -x -q -k "launch_order" > gpurun_out/r3_lo.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3_lo.log
tail -3 gpurun_out/r3_lo.log
A="--model hh --batch 393216 --nt 20001 --reps 1"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR --output-format csv -d $O/cf5a -- python3 tools/bench_closed_form.py $A > /dev/null 2> $O/cf5a.err || exit 1
rocprofv3 --pmc GRBM_GUI_ACTIVE FETCH_SIZE --output-format csv -d $O/cf5b -- python3 tools/bench_closed_form.py $A > /dev/null 2> $O/cf5b.err || exit 1
rocprofv3 --pmc WRITE_SIZE TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d $O/cf5c -- python3 tools/bench_closed_form.py $A > /dev/null 2> $O/cf5c.err || exit 1
for p in cf5a cf5b cf5c; do python3 tools/pmc_summary.py $O/$p > $O/$p.json; done
find $O -name "*counter_collection.csv" -delete
cat $O/cf5b.json
