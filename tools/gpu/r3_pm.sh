cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
for a in "--model hh --batch 393216" "--model hh --batch 393216 --protocol-major" "--model m6 --batch 65536" "--model m6 --batch 65536 --protocol-major" "--model nnf --batch 262144" "--model nnf --batch 262144 --protocol-major" "--model nnf --batch 65536" "--model nnf --batch 65536 --protocol-major"; do
  python3 tools/bench_closed_form.py --nt 20001 --reps 2 $a 2>/dev/null | python3 -c "
import sys,json
r=json.load(sys.stdin); print('$a', r['kernel'][-28:], round(r['ms'],2), round(r['frac_of_8TBps'],4), r['ok'])"
done
