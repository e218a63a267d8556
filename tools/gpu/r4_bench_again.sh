# round 4: the bench line again now that profiles/r04_pmc_summary.json carries this library's digest (roofline.traffic), then the crossovers
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04_prof
timeout -k 10 900 python3 bench.py > gpurun_out/r04_prof/r04_bench.json 2> gpurun_out/r04_prof/r04_bench.err || { tail -5 gpurun_out/r04_prof/r04_bench.err; exit 1; }
head -c 300 gpurun_out/r04_prof/r04_bench.json; echo
bash tools/gpu/r4_thresholds2.sh
