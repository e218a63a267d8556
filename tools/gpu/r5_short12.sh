#!/bin/bash
# round 5: the asm tile's two-MFMA form of k-tile 12 -- parity of the N = 200 tiles, then the headline command
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 800 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_round3.py tests/test_gpu_round4.py tests/test_gpu_edge.py tests/test_gpu_fuzz.py tests/test_step_traces.py tests/test_gpu_kats.py -x -q -m gpu > gpurun_out/r5_short12_tests.log 2>&1; rc=$?
tail -4 gpurun_out/r5_short12_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra-legs > gpurun_out/r5_short12_bench.json 2>/dev/null || exit 1
python3 -c "
import json; b=json.load(open('gpurun_out/r5_short12_bench.json')); print(b['value'], b['ms_per_step'], b['roofline']['frac'], b['roofline']['kernel_ms'])"
