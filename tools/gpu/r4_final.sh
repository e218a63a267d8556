# round 4: what the driver runs at round end -- the -m gpu suite, smoke(), the default bench line (timed)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/r4_tests.log 2>&1
rc=$?
tail -2 gpurun_out/r4_tests.log | cut -c1-300
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
SECONDS=0
timeout -k 10 900 python3 bench.py > gpurun_out/r4_bench_default.json 2> gpurun_out/r4_bench_default.err || { tail -5 gpurun_out/r4_bench_default.err; exit 1; }
echo "bench.py default run: ${SECONDS}s"
head -c 700 gpurun_out/r4_bench_default.json; echo
