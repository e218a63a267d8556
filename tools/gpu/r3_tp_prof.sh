cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3_tp_prof; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 tools/bench_grad.py --reps 1 > $O/out.json 2> $O/err.log || exit 1
find $O/trace -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
find $O -name "*kernel_trace.csv" -delete
grep ionode $O/kernel_stats.csv | cut -c1-200
tail -1 $O/out.json | cut -c1-300
