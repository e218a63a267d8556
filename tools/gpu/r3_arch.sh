cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
while IFS= read -r C; do
  [ -z "$C" ] && continue
  timeout -k 10 200 python3 tools/bench_closed_form.py $C --nt 20001 --reps 2 2>&1 | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$C', d['kernel'], 'ms %.2f'%d['ms'], 'us/eval(max) %.2f'%(d['ms']*1e3/d['max_nfe']), 'mean/max nfe %.3f'%(d['mean_nfe']/d['max_nfe']))"
done < tools/gpu/cases.txt
