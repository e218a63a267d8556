cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3_xcd; rm -rf $O; mkdir -p $O
for C in "--model hh --batch 393216" "--model hh --batch 393216 --index-order" "--model m6 --batch 65536" "--model nnf --batch 262144"; do
  timeout -k 10 200 python3 tools/bench_closed_form.py $C --nt 20001 --reps 3 2>&1 | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$C', d['kernel'][20:], 'ms %.2f'%d['ms'], 'frac %.4f'%d['frac_of_8TBps'])"
done
rocprofv3 --pmc GRBM_GUI_ACTIVE FETCH_SIZE --output-format csv -d $O/b -- python3 tools/bench_closed_form.py --model hh --batch 393216 --nt 20001 --reps 1 > /dev/null 2> $O/b.err || exit 1
python3 tools/pmc_summary.py $O/b
find $O -name "*counter_collection.csv" -delete
