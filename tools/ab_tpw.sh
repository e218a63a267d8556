cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
for a in "--model m6 --batch 32768" "--model m6 --batch 65536" "--model m6 --batch 131072" "--model m6 --batch 262144" "--batch 32768" "--batch 65536" "--batch 131072" "--batch 98304"; do
 for t in 16 64; do
  python3 tools/bench_closed_form.py --nt 20001 --reps 2 $a --tpw $t 2>/dev/null | python3 -c "
import sys,json
r=json.load(sys.stdin); print('$a tpw $t', r['kernel'][-28:], round(r['ms'],2), round(r['traj_per_s']/1e6,3))"
 done
done
