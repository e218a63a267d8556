#!/bin/bash
# round 5: the default bench line once more, with this library's counter summary in profiles/ (traffic, hbm_side counters filled in)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 400 python3 bench.py > gpurun_out/r05_bench_final.json 2> gpurun_out/r05_bench_final.err || exit 1
python3 -c "
import json; b=json.load(open('gpurun_out/r05_bench_final.json')); r=b['roofline']
print(b['value'], r['frac'], r.get('traffic'), r.get('mfma_busy'), r.get('useful_over_executed'))
print(json.dumps(r['hbm_side'])[:1500])"
