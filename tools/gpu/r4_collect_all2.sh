# round 4, final: evidence part 2 (lane-wise kernels' counters), then the bench line again so that roofline.traffic carries this library's counters
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
R=r04 PART=2 bash tools/collect_profiles.sh || exit 1
python3 tools/copy_profiles.py r04 > /dev/null 2>&1
timeout -k 10 900 python3 bench.py > gpurun_out/r04_prof/r04_bench.json 2> gpurun_out/r04_prof/r04_bench.err || { tail -5 gpurun_out/r04_prof/r04_bench.err; exit 1; }
python3 -c "
import json
d=json.loads(open('gpurun_out/r04_prof/r04_bench.json').read().strip().splitlines()[-1]); print(d['value'], d['roofline'])"
