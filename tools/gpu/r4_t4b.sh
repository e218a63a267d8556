cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_gpu_round4.py tests/test_gpu_round2.py::test_config1_single_sinewave_trajectory_through_the_shim -x -q -s > gpurun_out/r4_t4.log 2>&1
rc=$?
tail -6 gpurun_out/r4_t4.log | cut -c1-300
cat gpurun_out/config1_latency.json; echo
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python3 tools/bench_grad.py --reps 1 --budget-gb 64 > gpurun_out/r4_grad.json 2> gpurun_out/r4_grad.err; tail -1 gpurun_out/r4_grad.json | cut -c1-400
{
for a in "--model nnf --batch 16384 --tpw 64" "--model nnf --batch 16384 --tpw 1" "--model nnf --batch 32768 --tpw 64" "--model nnf --batch 32768 --tpw 1" "--model nnf --batch 49152 --tpw 64" "--model nnf --batch 49152 --tpw 1" "--model nnf --batch 65536 --tpw 64" "--model nnf --batch 65536 --tpw 1" "--model hh --batch 32768 --tpw 64" "--model hh --batch 32768 --tpw 16" "--model hh --batch 65536 --tpw 64" "--model hh --batch 65536 --tpw 16" "--model hh --batch 131072" "--model hh --batch 262144" "--model m6 --batch 16384 --tpw 64" "--model m6 --batch 16384 --tpw 16"  "--model m6 --batch 32768 --tpw 64" "--model m6 --batch 32768 --tpw 16" "--model m6 --batch 131072"; do
    timeout -k 10 200 python3 tools/bench_closed_form.py --nt 20001 --reps 3 $a 2>/dev/null | python3 -c "
import sys,json
r=json.load(sys.stdin); print('$a', r['kernel'][-28:], round(r['ms'],2), round(r['frac_of_8TBps'],4), r['ok'])"
done
} > gpurun_out/r4_thresholds.log 2>&1
cat gpurun_out/r4_thresholds.log
