# round 4: headline + other architectures after the output layer's batched LDS reads; tile parity subset first
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_edge.py -x -q > gpurun_out/r4_head_tests.log 2>&1 || { tail -5 gpurun_out/r4_head_tests.log; exit 1; }
tail -1 gpurun_out/r4_head_tests.log
timeout -k 10 600 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r4_head.json 2> gpurun_out/r4_head.err || { tail -5 gpurun_out/r4_head.err; exit 1; }
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r4_head.json').read().strip().splitlines()[-1])
print('headline', d['value'], d['roofline']['frac'], d['roofline']['kernel_ms'])
for k,v in d['other_architectures_4096'].items(): print(k, round(v['kernel_ms'],2), round(v['frac'],4))
print('order', {k:(round(v,4) if isinstance(v,float) else v) for k,v in d['launch_order_16384'].items() if 'frac' in k or 'order' in k})
print('c3', d['config3_nnd_staircase_16384'].get('frac'), d['config3_nnd_staircase_16384'].get('kernel_ms'))
print('lat', d['config1_latency']['ms_per_call'], 'grad', d['gradient_config5']['forward_with_checkpoints_s'], d['gradient_config5']['backward_s'], d['gradient_config5']['frac'])
PY
