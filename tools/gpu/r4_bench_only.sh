cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04_prof
timeout -k 10 900 python3 bench.py > gpurun_out/r04_prof/r04_bench.json 2> gpurun_out/r04_prof/r04_bench.err || { tail -5 gpurun_out/r04_prof/r04_bench.err; exit 1; }
python3 -c "
import json
d=json.loads(open('gpurun_out/r04_prof/r04_bench.json').read().strip().splitlines()[-1]); print(d['value'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline'].get('traffic_stale'))"
