#!/bin/bash
# Run ON THE GPU BOX (gpurun): round-2 evidence.  Every rocprofv3 run puts the program itself after `--`; counters are
# collected in their own passes (no --kernel-trace with --pmc).  Results land under gpurun_out/r02_prof/.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02_prof; mkdir -p $O
python3 bench.py > $O/r02_bench.json 2> $O/r02_bench.err || exit 1
# headline command alone (the s00 kernel's average launch duration must agree with the bench line's roofline.kernel_ms) ...
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extra-legs > $O/r02_bench_profiled.json 2> $O/trace.err || exit 1
# ... and with the extra legs (closed-form, gradient, regression, launch order: the last re-uses the s00 kernel at 16384 trajectories)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_legs -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/r02_bench_profiled_legs.json 2> $O/trace_legs.err || exit 1
S="--steps 1 --warmup 0 --no-cpu-baseline --no-extra-legs"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc1 -- python3 bench.py $S > /dev/null 2> $O/pmc1.err || exit 1
rocprofv3 --pmc GRBM_GUI_ACTIVE FETCH_SIZE --output-format csv -d $O/pmc2 -- python3 bench.py $S > /dev/null 2> $O/pmc2.err || exit 1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc3 -- python3 bench.py $S > /dev/null 2> $O/pmc3.err || exit 1
C="--batch 262144 --nt 20001 --reps 1"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR --output-format csv -d $O/cf1 -- python3 tools/bench_closed_form.py $C > /dev/null 2> $O/cf1.err || exit 1
rocprofv3 --pmc GRBM_GUI_ACTIVE FETCH_SIZE --output-format csv -d $O/cf2 -- python3 tools/bench_closed_form.py $C > /dev/null 2> $O/cf2.err || exit 1
rocprofv3 --pmc WRITE_SIZE TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d $O/cf3 -- python3 tools/bench_closed_form.py $C > /dev/null 2> $O/cf3.err || exit 1
for p in pmc1 pmc2 pmc3 cf1 cf2 cf3; do python3 tools/pmc_summary.py $O/$p > $O/$p.json; done
find $O/trace -name "*kernel_stats.csv" -exec cp {} $O/r02_kernel_stats.csv \;
find $O/trace_legs -name "*kernel_stats.csv" -exec cp {} $O/r02_kernel_stats_legs.csv \;
# keep the merge small: drop the raw per-dispatch traces
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete
head -c 1500 $O/r02_bench.json; echo; head -6 $O/r02_kernel_stats.csv | cut -c1-200
