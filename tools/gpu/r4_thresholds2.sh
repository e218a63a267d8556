# round 4: trajectories-per-wavefront crossovers after the kernel changes of this round (16 vs 64 per wavefront)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
{
for m in "nnf:1" "hh:16" "m6:16"; do
  model=${m%%:*}; small=${m##*:}
  for B in 8192 16384 24576 32768 49152 65536; do
    for tpw in $small 64; do
      timeout -k 10 200 python3 tools/bench_closed_form.py --model $model --batch $B --nt 20001 --reps 3 --tpw $tpw 2>/dev/null | python3 -c "
import sys,json
r=json.load(sys.stdin); print('$model B=$B tpw=$tpw', r['kernel'][-28:], round(r['ms'],2), r['ok'])"
    done
  done
done
} > gpurun_out/r4_thresholds2.log 2>&1
cat gpurun_out/r4_thresholds2.log
