cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
for a in "--model nnf --batch 65536" "--model nnf --batch 65536 --tpw 64" "--model nnf --batch 131072" "--model nnf --batch 131072 --tpw 64" "--model nnf --batch 65536 --layers 1" "--model nnf --batch 65536 --layers 1 --tpw 64"; do
  python3 tools/bench_closed_form.py --nt 20001 --reps 2 $a 2>/dev/null | python3 -c "
import sys,json
r=json.load(sys.stdin); print('$a', r['kernel'][-28:], round(r['ms'],2), round(r['frac_of_8TBps'],4), r['ok'], r['mean_nfe'])"
done
