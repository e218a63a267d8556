cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3_icache; mkdir -p $O
rocprofv3 -L > $O/counters.txt 2>&1
grep -i "icache\|SQC_\|IFETCH\|INST_CACHE" $O/counters.txt | head -40
S="--steps 1 --warmup 0 --no-cpu-baseline --no-extra-legs"
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_IFETCH --output-format csv -d $O/pmc -- python3 bench.py $S > /dev/null 2> $O/pmc.err
python3 tools/pmc_summary.py $O/pmc | head -60
tail -3 $O/pmc.err
