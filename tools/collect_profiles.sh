#!/bin/bash
# Run ON THE GPU BOX (gpurun): the round's evidence.  Every rocprofv3 run puts the program itself after `--`; counters are
# collected in their own passes (no --kernel-trace with --pmc).  Results land under gpurun_out/${R}_prof/ (R = r03 by default);
# tools/copy_profiles.py copies the summaries into profiles/ (tracked).  Usage on the box: R=r05 bash tools/collect_profiles.sh
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
R=${R:-r05}
O=gpurun_out/${R}_prof; mkdir -p $O
python3 -c "import importlib,sys; sys.path.insert(0,'.'); print(importlib.import_module('neural-ode-ion-channels_amd').capi.library_digest())" > $O/libionode.sha256
# two halves (a gpurun call is limited to 20 minutes): PART=1 bench line + kernel traces + headline counters, PART=2 the lane-wise
# kernels' counters + summaries; no PART: everything
PART=${PART:-12}
if [[ $PART == *1* ]]; then
python3 bench.py > $O/${R}_bench.json 2> $O/${R}_bench.err || exit 1
# headline command alone (the s00 kernel's average launch duration must agree with the bench line's roofline.kernel_ms) ...
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extra-legs > $O/${R}_bench_profiled.json 2> $O/trace.err || exit 1
# ... and with the extra legs (closed-form, gradient, regression, launch order: the last re-uses the s00 kernel at 16384 trajectories)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_legs -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/${R}_bench_profiled_legs.json 2> $O/trace_legs.err || exit 1
S="--steps 1 --warmup 0 --no-cpu-baseline --no-extra-legs"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc1 -- python3 bench.py $S > /dev/null 2> $O/pmc1.err || exit 1
rocprofv3 --pmc GRBM_GUI_ACTIVE FETCH_SIZE --output-format csv -d $O/pmc2 -- python3 bench.py $S > /dev/null 2> $O/pmc2.err || exit 1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc3 -- python3 bench.py $S > /dev/null 2> $O/pmc3.err || exit 1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM --output-format csv -d $O/pmc4 -- python3 bench.py $S > /dev/null 2> $O/pmc4.err || exit 1
fi
if [[ $PART == *2* ]]; then
# the HBM-side kernels: HH 2-state (two residency rounds), 6-state, both N <= 16 kernels (16 / 64 trajectories per wavefront),
# HH again with the library's default (protocol-major) launch order, and the 6-state two-per-SIMD build (262144)
i=0
# (cf7, cf8 since round 5: the 2-state kernel at two full residency rounds, and the 5 x 10 net in the reference's fp32 state -- every
# entry of bench.py's roofline.hbm_side has its counters)
for C in "--model hh --batch 393216 --index-order" "--model m6 --batch 65536" "--model nnf --batch 65536" "--model nnf --batch 262144" "--model hh --batch 393216" "--model m6 --batch 262144" "--model hh --batch 524288" "--model nnf --batch 262144 --f32"; do
  i=$((i+1)); A="$C --nt 20001 --reps 1"
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d $O/cf${i}a -- python3 tools/bench_closed_form.py $A > /dev/null 2> $O/cf${i}a.err || exit 1
  rocprofv3 --pmc GRBM_GUI_ACTIVE FETCH_SIZE --output-format csv -d $O/cf${i}b -- python3 tools/bench_closed_form.py $A > /dev/null 2> $O/cf${i}b.err || exit 1
  rocprofv3 --pmc WRITE_SIZE TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/cf${i}c -- python3 tools/bench_closed_form.py $A > /dev/null 2> $O/cf${i}c.err || exit 1
done
fi
if [[ $PART == *3* ]]; then
# round 5: the regression step's three kernels (tile kernel, reduce, Adam): per-kernel time, then MFMA / VALU / LDS / L2 counters
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_reg -- python3 tools/bench_regression.py --iters 20 > $O/${R}_regression.json 2> $O/trace_reg.err || exit 1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_LDS --output-format csv -d $O/reg1 -- python3 tools/bench_regression.py --iters 2 > /dev/null 2> $O/reg1.err || exit 1
rocprofv3 --pmc GRBM_GUI_ACTIVE FETCH_SIZE --output-format csv -d $O/reg2 -- python3 tools/bench_regression.py --iters 2 > /dev/null 2> $O/reg2.err || exit 1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/reg3 -- python3 tools/bench_regression.py --iters 2 > /dev/null 2> $O/reg3.err || exit 1
[ -d $O/trace_reg ] && find $O/trace_reg -name "*kernel_stats.csv" -exec cp {} $O/${R}_kernel_stats_regression.csv \;
fi
if [[ $PART == *4* ]]; then
# round 5: the gradient share's kernels (forward with checkpoints, recompute, walk, reduce): per-kernel time, MFMA / VALU counters
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_grad -- python3 tools/bench_grad.py --reps 1 > $O/${R}_gradient.json 2> $O/trace_grad.err || exit 1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_LDS --output-format csv -d $O/grad1 -- python3 tools/bench_grad.py --reps 1 > /dev/null 2> $O/grad1.err || exit 1
rocprofv3 --pmc GRBM_GUI_ACTIVE FETCH_SIZE --output-format csv -d $O/grad2 -- python3 tools/bench_grad.py --reps 1 > /dev/null 2> $O/grad2.err || exit 1
[ -d $O/trace_grad ] && find $O/trace_grad -name "*kernel_stats.csv" -exec cp {} $O/${R}_kernel_stats_gradient.csv \;
fi
for p in grad1 grad2 reg1 reg2 reg3 pmc1 pmc2 pmc3 pmc4 cf1a cf1b cf1c cf2a cf2b cf2c cf3a cf3b cf3c cf4a cf4b cf4c cf5a cf5b cf5c cf6a cf6b cf6c cf7a cf7b cf7c cf8a cf8b cf8c; do [ -d $O/$p ] && python3 tools/pmc_summary.py $O/$p > $O/$p.json; done
[ -d $O/trace ] && find $O/trace -name "*kernel_stats.csv" -exec cp {} $O/${R}_kernel_stats.csv \;
[ -d $O/trace_legs ] && find $O/trace_legs -name "*kernel_stats.csv" -exec cp {} $O/${R}_kernel_stats_legs.csv \;
# keep the merge small: drop the raw per-dispatch traces
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete
[ -f $O/${R}_bench.json ] && head -c 1500 $O/${R}_bench.json; echo; [ -f $O/${R}_kernel_stats.csv ] && head -6 $O/${R}_kernel_stats.csv | cut -c1-200; true
